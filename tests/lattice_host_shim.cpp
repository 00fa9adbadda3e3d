// host build of libecc_amd/csrc/ecamd_lattice.h for tests/test_lattice_host.py (test infrastructure)
#include "../libecc_amd/csrc/ecamd_lattice.h"
extern "C" int lat_reduce_host(const uint32_t *q, const uint32_t *h, uint32_t *v, uint32_t *u, int *neg, int *iters)
{
	bool n = false;
	const bool ok = lat_reduce(q, h, v, u, &n, iters);
	*neg = n ? 1 : 0;
	return ok ? 1 : 0;
}
