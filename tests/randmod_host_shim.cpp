// host build of libecc_amd/csrc/ecamd_randmod.h for tests/test_randmod_host.py (test infrastructure)
#include "../libecc_amd/csrc/ecamd_randmod.h"
extern "C" int randmod_host(int nw, uint32_t *out, const uint8_t *raw, int rawlen, const uint32_t *q)
{
	switch (nw) {
	case 6: randmod_words<6>(out, raw, rawlen, q); return 0;
	case 7: randmod_words<7>(out, raw, rawlen, q); return 0;
	case 8: randmod_words<8>(out, raw, rawlen, q); return 0;
	case 12: randmod_words<12>(out, raw, rawlen, q); return 0;
	case 17: randmod_words<17>(out, raw, rawlen, q); return 0;
	default: return -1;
	}
}
