"""ctypes binding of include/libecc_amd.h (one Python method per C entry point)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ECAMD_OK, ECAMD_ERR, ECAMD_INF = 0, 1, 2
FP_MUL_MONTY, FP_ADD, FP_SUB, FP_MUL, FP_INV = 0, 1, 2, 3, 4

# every symbol include/libecc_amd.h declares (tests check the .so exports exactly these)
EXPORTED_SYMBOLS = [
    "ecamd_device_count", "ecamd_ctx_create", "ecamd_ctx_destroy", "ecamd_last_error",
    "ecamd_ctx_set_max_chunk", "ecamd_ctx_set_secret_scalars", "ecamd_ctx_set_eddsa_msm", "ec_eddsa_verify_all_batch_dev", "ecamd_debug_eddsa_msm", "ec_eddsa_encode_point_batch", "ecamd_multi_eddsa_encode_point_batch", "ecamd_ctx_enable_kernel_timing", "ecamd_ctx_kernel_times", "ecamd_curve_by_name", "ecamd_curve_from_params", "ecamd_curve_free",
    "ecamd_curve_coord_len", "ecamd_curve_order_len", "ecamd_curve_words", "ec_prj_pt_mul_batch", "ec_prj_pt_mul_blind_batch",
    "ec_prj_pt_mul_batch_dev", "ecamd_ctx_synchronize", "ec_prj_pt_add_batch", "ec_prj_pt_dbl_batch",
    "ec_fp_op_batch", "ec_ecdsa_verify_batch", "ec_ecdsa_verify_batch_fmt", "ec_ecdsa_sign_batch", "ec_ecccdh_derive_batch", "ec_xdh_batch",
    "ec_prj_pt_mul_batch_fmt", "ec_prj_pt_unique_batch", "ec_structured_pub_key_import_batch",
    "ec_ecdsa_verify_msg_batch_fmt", "ec_eddsa_verify_msg_batch", "ecamd_multi_ecdsa_verify_msg_batch_fmt", "ecamd_multi_eddsa_verify_msg_batch",
    "ec_prj_pt_op_batch_fmt", "ec_prj_pt_unprotected_mult_batch", "ecamd_multi_prj_pt_op_batch_fmt", "ecamd_multi_prj_pt_unprotected_mult_batch",
    "ec_aff_pt_y_from_x_batch", "ec_point_decompress_batch", "ec_structured_sig_import_batch", "ec_structured_key_pair_import_batch",
    "ec_eddsa_verify_batch", "ec_eddsa_verify_all_batch", "ec_ecdsa_verify_batch_dev", "ec_eddsa_verify_batch_dev", "ec_xdh_batch_dev",
    "ec_ecdsa_sign_batch_dev", "ec_ecccdh_derive_batch_dev", "ec_eddsa_sign_R_batch", "ec_eddsa_sign_S_batch",
    "ecamd_multi_create", "ecamd_multi_destroy", "ecamd_multi_size", "ecamd_multi_device", "ecamd_multi_ctx",
    "ecamd_multi_shard_range", "ecamd_multi_curve_by_name", "ecamd_multi_curve_from_params", "ecamd_multi_curve_free",
    "ecamd_multi_curve_handle", "ecamd_multi_curve_coord_len", "ecamd_multi_curve_order_len",
    "ecamd_multi_prj_pt_mul_batch", "ecamd_multi_prj_pt_mul_batch_fmt", "ecamd_multi_prj_pt_unique_batch",
    "ecamd_multi_ecdsa_verify_batch", "ecamd_multi_ecdsa_verify_batch_fmt", "ecamd_multi_ecdsa_sign_batch", "ecamd_multi_ecccdh_derive_batch",
    "ecamd_multi_xdh_batch", "ecamd_multi_eddsa_verify_batch", "ecamd_multi_eddsa_verify_all_batch", "ecamd_multi_allgather",
    "ecamd_multi_allgather_streams", "ecamd_multi_eddsa_sign_R_batch", "ecamd_multi_eddsa_sign_S_batch",
    "ecamd_multi_set_secret_scalars", "ecamd_multi_wipe_scratch", "ecamd_ctx_wipe_scratch", "ecamd_ctx_stream", "ecamd_host_alloc", "ecamd_host_free", "ecamd_ctx_dominant_kernel_ms", "ecamd_ctx_set_msm_seed", "ecamd_ctx_discard_msm_seed", "ecamd_multi_set_msm_seed", "ec_eddsa_verify_ph_prj_batch", "ecamd_multi_eddsa_verify_ph_prj_batch", "ec_nn_random_mod_batch", "ec_ecdsa_sign_msg_batch", "ec_key_pair_gen_raw_batch", "ecamd_multi_ecdsa_sign_msg_batch", "ecamd_multi_key_pair_gen_raw_batch", "ec_eddsa_verify_msg_prj_batch", "ecamd_multi_eddsa_verify_msg_prj_batch", "ecamd_ctx_set_host_ready_hook", "ecamd_multi_set_host_ready_hook", "ecamd_multi_prj_pt_add_batch",
    "ec_schnorr_verify_all_batch", "ec_schnorr_verify_all_batch_dev", "ec_schnorr_verify_msg_all_batch", "ec_eddsa_verify_msg_prj_all_batch", "ecamd_multi_eddsa_verify_msg_prj_all_batch", "ecamd_multi_schnorr_verify_msg_all_batch", "ec_schnorr_verify_all_available", "ecamd_multi_schnorr_verify_all_batch", "ecamd_debug_schnorr_msm", "ecamd_debug_schnorr_msm_words",
]


class EcamdError(RuntimeError):
    pass


def lib_path():
    # ECAMD_LIB_PATH: developer override to A/B-test an alternative build of the same library
    return os.environ.get("ECAMD_LIB_PATH") or os.path.join(HERE, "lib", "libecc_amd.so")


_LIB = None


def load_library():
    """Load the HIP shared library; raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise EcamdError(f"{path} is missing: run `python libecc_amd/build.py` "
                             "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(path)
        vp, u8p, u32 = C.c_void_p, C.c_char_p, C.c_uint32
        L.ecamd_last_error.restype = C.c_char_p
        L.ecamd_ctx_create.argtypes = [C.POINTER(vp), C.c_int]
        L.ecamd_ctx_destroy.argtypes = [vp]
        L.ecamd_ctx_destroy.restype = None
        L.ecamd_ctx_set_max_chunk.argtypes = [vp, u32]
        L.ecamd_ctx_set_secret_scalars.argtypes = [vp, C.c_int]
        L.ecamd_ctx_synchronize.argtypes = [vp]
        L.ecamd_ctx_enable_kernel_timing.argtypes = [vp, C.c_int]
        L.ecamd_ctx_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.c_int]
        L.ecamd_curve_by_name.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        L.ecamd_curve_from_params.argtypes = [vp] + [u8p, u32] * 7 + [C.POINTER(vp)]
        L.ecamd_curve_free.argtypes = [vp]
        L.ecamd_curve_free.restype = None
        for f in ("ecamd_curve_coord_len", "ecamd_curve_order_len", "ecamd_curve_words"):
            getattr(L, f).argtypes = [vp]
        L.ec_prj_pt_mul_batch.argtypes = [vp, vp, u32, u8p, u32, u8p, u8p, u8p]
        L.ec_prj_pt_mul_blind_batch.argtypes = [vp, vp, u32, u8p, u32, u8p, u32, u8p, u8p, u8p]
        L.ec_prj_pt_mul_batch_dev.argtypes = [vp, vp, u32, vp, u32, vp, vp, vp, vp]
        L.ec_prj_pt_add_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ec_prj_pt_dbl_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p]
        L.ec_fp_op_batch.argtypes = [vp, vp, C.c_int, u32, vp, vp, vp]
        L.ec_ecdsa_verify_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p]
        L.ec_ecdsa_sign_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p, u8p]
        L.ec_ecdsa_verify_batch_fmt.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, u8p, u32, u8p]
        L.ec_xdh_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ec_eddsa_verify_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p]
        L.ec_eddsa_verify_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.ecamd_debug_eddsa_msm.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p, C.POINTER(C.c_int), vp, vp]
        L.ecamd_ctx_set_eddsa_msm.argtypes = [vp, C.c_int, u32, u32]
        L.ec_schnorr_verify_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p, C.c_int, C.POINTER(C.c_int)]
        L.ec_schnorr_verify_all_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp, C.c_int, vp, vp]
        L.ec_eddsa_verify_msg_prj_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u32, C.POINTER(C.c_int)]
        L.ecamd_multi_eddsa_verify_msg_prj_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u32, C.POINTER(C.c_int)]
        L.ec_schnorr_verify_msg_all_batch.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, u32, u32, C.POINTER(C.c_int)]
        L.ecamd_multi_schnorr_verify_msg_all_batch.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, C.c_int, C.c_int, u8p, u32, u32, C.POINTER(C.c_int)]
        L.ecamd_multi_schnorr_verify_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p, C.c_int, C.POINTER(C.c_int)]
        L.ec_schnorr_verify_all_available.argtypes = [vp, C.c_int]
        L.ecamd_debug_schnorr_msm.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p, C.c_int, u8p, C.POINTER(C.c_int), vp, vp]
        L.ecamd_debug_schnorr_msm_words.argtypes = [vp]
        L.ecamd_debug_schnorr_msm_words.restype = C.c_uint32
        L.ec_eddsa_verify_all_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp]
        L.ec_eddsa_encode_point_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p]
        L.ecamd_multi_eddsa_encode_point_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p]
        L.ec_prj_pt_mul_batch_fmt.argtypes = [vp, vp, u32, u8p, u32, u8p, C.c_int, u8p, C.c_int, u8p]
        L.ec_prj_pt_unique_batch.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, C.c_int, u8p]
        L.ec_structured_pub_key_import_batch.argtypes = [vp, vp, u32, u8p, u32, C.c_int, u8p, u8p]
        L.ec_aff_pt_y_from_x_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ec_point_decompress_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p]
        L.ec_structured_sig_import_batch.argtypes = [vp, u32, u8p, u32, C.c_int, C.c_int, u8p, u8p]
        L.ec_structured_key_pair_import_batch.argtypes = [vp, vp, u32, u8p, u32, C.c_int, u8p, u8p, u8p]
        L.ec_ecdsa_verify_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp]
        L.ec_eddsa_verify_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp]
        L.ec_eddsa_sign_R_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p]
        L.ec_eddsa_sign_S_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ec_ecdsa_sign_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, u32, vp, vp, vp]
        L.ec_ecccdh_derive_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp]
        L.ec_xdh_batch_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp]
        L.ec_ecccdh_derive_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        # several GPUs from C
        L.ecamd_multi_create.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
        L.ecamd_multi_destroy.argtypes = [vp]
        L.ecamd_multi_destroy.restype = None
        L.ecamd_multi_size.argtypes = [vp]
        L.ecamd_multi_device.argtypes = [vp, C.c_int]
        L.ecamd_multi_ctx.argtypes = [vp, C.c_int]
        L.ecamd_multi_ctx.restype = vp
        L.ecamd_multi_shard_range.argtypes = [u32, C.c_int, C.c_int, C.POINTER(u32), C.POINTER(u32)]
        L.ecamd_multi_shard_range.restype = None
        L.ecamd_multi_curve_by_name.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
        L.ecamd_multi_curve_from_params.argtypes = [vp] + [u8p, u32] * 7 + [C.POINTER(vp)]
        L.ecamd_multi_curve_free.argtypes = [vp]
        L.ecamd_multi_curve_free.restype = None
        L.ecamd_multi_curve_handle.argtypes = [vp, C.c_int]
        L.ecamd_multi_curve_handle.restype = vp
        L.ecamd_multi_curve_coord_len.argtypes = [vp]
        L.ecamd_multi_curve_order_len.argtypes = [vp]
        L.ecamd_multi_prj_pt_mul_batch.argtypes = [vp, vp, u32, u8p, u32, u8p, u8p, u8p]
        L.ecamd_multi_prj_pt_mul_batch_fmt.argtypes = [vp, vp, u32, u8p, u32, u8p, C.c_int, u8p, C.c_int, u8p]
        L.ecamd_multi_prj_pt_unique_batch.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, C.c_int, u8p]
        L.ecamd_multi_ecdsa_verify_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p]
        L.ecamd_multi_ecdsa_verify_batch_fmt.argtypes = [vp, vp, u32, u8p, C.c_int, u8p, u8p, u32, u8p]
        L.ecamd_multi_ecdsa_sign_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p, u8p]
        L.ecamd_multi_ecccdh_derive_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ecamd_multi_xdh_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u8p]
        L.ecamd_multi_eddsa_verify_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, u8p]
        L.ecamd_multi_eddsa_verify_all_batch.argtypes = [vp, vp, u32, u8p, u8p, u8p, u32, C.POINTER(C.c_int), C.POINTER(C.c_uint32)]
        L.ecamd_multi_allgather.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.c_size_t]
        _LIB = L
    return _LIB


def _chk(L, ret, what):
    if ret != 0:
        raise EcamdError(f"{what}: {L.ecamd_last_error().decode()}")


class Context:
    """ecamd_ctx: one GPU."""

    def __init__(self, device=0):
        self.L = load_library()
        self.h = C.c_void_p()
        _chk(self.L, self.L.ecamd_ctx_create(C.byref(self.h), device), "ecamd_ctx_create")

    def close(self):
        if self.h:
            self.L.ecamd_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def set_max_chunk(self, n):
        _chk(self.L, self.L.ecamd_ctx_set_max_chunk(self.h, n), "ecamd_ctx_set_max_chunk")

    def set_secret_scalars(self, on=True):
        _chk(self.L, self.L.ecamd_ctx_set_secret_scalars(self.h, 1 if on else 0), "ecamd_ctx_set_secret_scalars")

    def set_eddsa_msm(self, mode=1, min_items=0, items_per_lane=0):
        """Ed25519 whole-batch verification through the multi-scalar multiplication: 0 never, 1 large batches, 2 always"""
        _chk(self.L, self.L.ecamd_ctx_set_eddsa_msm(self.h, mode, min_items, items_per_lane), "ecamd_ctx_set_eddsa_msm")

    def set_host_ready_hook(self, fn):
        """ecamd_ctx_set_host_ready_hook: fn(first, count) is called before the host-pointer entry points read that range of their
        input arrays (None clears it)"""
        HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)
        self._hook = HOOK((lambda arg, first, count: fn(first, count))) if fn else C.cast(None, HOOK)   # kept alive with the context
        self.L.ecamd_ctx_set_host_ready_hook.argtypes = [C.c_void_p, HOOK, C.c_void_p]
        self.L.ecamd_ctx_set_host_ready_hook.restype = C.c_int
        _chk(self.L, self.L.ecamd_ctx_set_host_ready_hook(self.h, self._hook, None), "ecamd_ctx_set_host_ready_hook")

    def enable_kernel_timing(self, on=True):
        _chk(self.L, self.L.ecamd_ctx_enable_kernel_timing(self.h, 1 if on else 0), "ecamd_ctx_enable_kernel_timing")

    def kernel_times(self):
        """ms of [table, affine, loop, finalize] of the last fast-path batch (HIP events on its stream)"""
        ms = (C.c_double * 4)()
        _chk(self.L, self.L.ecamd_ctx_kernel_times(self.h, ms, 4), "ecamd_ctx_kernel_times")
        return list(ms)

    def dominant_kernel_ms(self):
        """ms of the dominant kernel of the last protocol call (verify loop / ladder / Edwards window loop), timing enabled"""
        ms = C.c_double(0.0)
        _chk(self.L, self.L.ecamd_ctx_dominant_kernel_ms(self.h, C.byref(ms)), "ecamd_ctx_dominant_kernel_ms")
        return ms.value

    def synchronize(self):
        _chk(self.L, self.L.ecamd_ctx_synchronize(self.h), "ecamd_ctx_synchronize")

    def curve(self, name):
        return Curve(self, name)


class Curve:
    """ecamd_curve: the ec_params equivalent, bound to a context."""

    def __init__(self, ctx, name=None, params=None):
        self.ctx, self.L = ctx, ctx.L
        self.h = C.c_void_p()
        if params is None:
            _chk(self.L, self.L.ecamd_curve_by_name(ctx.h, name.encode(), C.byref(self.h)),
                 "ecamd_curve_by_name")
        else:
            args = []
            for k in ("p", "a", "b", "order", "gx", "gy", "q"):
                v = params[k]
                b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
                args += [b, len(b)]
            _chk(self.L, self.L.ecamd_curve_from_params(ctx.h, *args, C.byref(self.h)),
                 "ecamd_curve_from_params")
        self.name = name
        self.clen = self.L.ecamd_curve_coord_len(self.h)
        self.qlen = self.L.ecamd_curve_order_len(self.h)
        self.words = self.L.ecamd_curve_words(self.h)

    def free(self):
        if self.h:
            self.L.ecamd_curve_free(self.h)
            self.h = C.c_void_p()

    # -- batched prj_pt_mul, host buffers (bytes in, bytes out) --
    def scalar_mult(self, scalars, points=None, slen=None):
        slen = slen or self.qlen
        n = len(scalars) // slen
        out = C.create_string_buffer(max(1, 2 * self.clen * n))
        st = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_mul_batch(self.ctx.h, self.h, n, scalars, slen, points, out, st),
             "ec_prj_pt_mul_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]

    def scalar_mult_blind(self, scalars, blinds, points=None, slen=None, blen=None):
        """prj_pt_mul_blind: the device multiplies by m + b #E"""
        slen, blen = slen or self.qlen, blen or self.qlen
        n = len(scalars) // slen
        out, st = C.create_string_buffer(max(1, 2 * self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_mul_blind_batch(self.ctx.h, self.h, n, scalars, slen, blinds, blen, points, out, st),
             "ec_prj_pt_mul_blind_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]

    # -- batched prj_pt_mul, device pointers (e.g. torch tensors' data_ptr()), asynchronous --
    def scalar_mult_dev(self, n, d_scalars, slen, d_points, d_out, d_status, stream=None):
        _chk(self.L, self.L.ec_prj_pt_mul_batch_dev(self.ctx.h, self.h, n, d_scalars, slen, d_points,
                                                     d_out, d_status, stream),
             "ec_prj_pt_mul_batch_dev")

    def pt_add(self, p1, p2=None):
        n = len(p1) // (2 * self.clen)
        out = C.create_string_buffer(max(1, 2 * self.clen * n))
        st = C.create_string_buffer(max(1, n))
        if p2 is None:
            _chk(self.L, self.L.ec_prj_pt_dbl_batch(self.ctx.h, self.h, n, p1, out, st), "ec_prj_pt_dbl_batch")
        else:
            _chk(self.L, self.L.ec_prj_pt_add_batch(self.ctx.h, self.h, n, p1, p2, out, st),
                 "ec_prj_pt_add_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]

    def fp_op(self, op, a, b):
        """a, b: lists of Python ints < p; returns a list of ints (libecc 64-bit limb layout inside)."""
        n = len(a)
        nl = (self.clen * 8 + 63) // 64
        mask = 2**64 - 1
        A = (C.c_uint64 * (n * nl))(*[(x >> (64 * k)) & mask for x in a for k in range(nl)])
        B = (C.c_uint64 * (n * nl))(*[(x >> (64 * k)) & mask for x in b for k in range(nl)])
        O = (C.c_uint64 * (n * nl))()
        _chk(self.L, self.L.ec_fp_op_batch(self.ctx.h, self.h, op, n, A, B, O), "ec_fp_op_batch")
        return [sum(O[i * nl + k] << (64 * k) for k in range(nl)) for i in range(n)]

    # -- protocol callers --
    def ecdsa_verify(self, pubs, sigs, digests, hlen):
        n = len(pubs) // (2 * self.clen)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecdsa_verify_batch(self.ctx.h, self.h, n, pubs, sigs, digests, hlen, res),
             "ec_ecdsa_verify_batch")
        return res.raw[:n]

    @staticmethod
    def msg_slots(msgs, stride=None):
        """fixed-stride slots (u32 little-endian length + bytes) of a list of messages, as the *_msg_* entry points take them"""
        stride = stride or ((4 + max([len(m) for m in msgs] + [0]) + 3) // 4) * 4
        buf = bytearray(stride * len(msgs))
        for i, m in enumerate(msgs):
            buf[stride * i:stride * i + 4] = len(m).to_bytes(4, "little")
            buf[stride * i + 4:stride * i + 4 + len(m)] = m
        return bytes(buf), stride

    def ecdsa_verify_msgs(self, pubs, pub_fmt, sigs, hash_type, msgs):
        """ECDSA verification with H(m) computed on the device (hash_type 1..4 = SHA-224/256/384/512); msgs: list of bytes"""
        n = len(msgs)
        slots, stride = self.msg_slots(msgs)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecdsa_verify_msg_batch_fmt(self.ctx.h, self.h, n, pubs, pub_fmt, sigs, hash_type, slots, stride, res),
             "ec_ecdsa_verify_msg_batch_fmt")
        return res.raw[:n]

    def eddsa_verify_msgs(self, pubkeys, sigs, hash_inputs):
        """Ed25519 verification with hram = SHA-512(dom2 || R || A || PH(M)) computed on the device; hash_inputs: list of bytes"""
        n = len(hash_inputs)
        slots, stride = self.msg_slots(hash_inputs)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_verify_msg_batch(self.ctx.h, self.h, n, pubkeys, sigs, slots, stride, res), "ec_eddsa_verify_msg_batch")
        return res.raw[:n]

    def eddsa_verify_msgs_prj(self, keys_prj, sigs, hash_inputs, a_offset):
        """the same from projective keys (X || Y || Z on WEI25519): the device writes each key's encoding into bytes
        a_offset .. a_offset + 32 of its hash input (left blank by the caller) before hashing"""
        n = len(hash_inputs)
        slots, stride = self.msg_slots(hash_inputs)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_verify_msg_prj_batch(self.ctx.h, self.h, n, keys_prj, sigs, slots, stride, a_offset, res),
             "ec_eddsa_verify_msg_prj_batch")
        return res.raw[:n]

    def eddsa_verify_ph_prj(self, keys_prj, sigs, hash_inputs, a_offset, msgs):
        """EDDSA25519PH from projective keys: hash_inputs hold dom2 || R || 96 blank octets (A, PH(M)) at a_offset; the device hashes
        msgs, fills both blanks, hashes again and verifies"""
        n = len(hash_inputs)
        slots, stride = self.msg_slots(hash_inputs)
        mslots, mstride = self.msg_slots(msgs)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_verify_ph_prj_batch(self.ctx.h, self.h, n, keys_prj, sigs, slots, stride, a_offset, mslots, mstride, res),
             "ec_eddsa_verify_ph_prj_batch")
        return res.raw[:n]

    def ecdsa_verify_fmt(self, pubs, pub_fmt, sigs, digests, hlen):
        n = len(pubs) // ((3 if pub_fmt else 2) * self.clen)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecdsa_verify_batch_fmt(self.ctx.h, self.h, n, pubs, pub_fmt, sigs, digests, hlen, res),
             "ec_ecdsa_verify_batch_fmt")
        return res.raw[:n]

    def ecdsa_sign(self, privs, nonces, digests, hlen):
        n = len(privs) // self.qlen
        sigs = C.create_string_buffer(max(1, 2 * self.qlen * n))
        st = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecdsa_sign_batch(self.ctx.h, self.h, n, privs, nonces, digests, hlen, sigs, st),
             "ec_ecdsa_sign_batch")
        return sigs.raw[:2 * self.qlen * n], st.raw[:n]

    def random_mod(self, raw):
        """nn_get_random_mod given its 2 * qlen random bytes per item: LE(raw) mod (q - 1) + 1, big-endian"""
        n = len(raw) // (2 * self.qlen)
        out = C.create_string_buffer(max(1, self.qlen * n))
        _chk(self.L, self.L.ec_nn_random_mod_batch(self.ctx.h, self.h, n, raw, out), "ec_nn_random_mod_batch")
        return out.raw[:self.qlen * n]

    def ecdsa_sign_msgs(self, privs, nonce_raw, hash_type, msgs, stride=None):
        """ECDSA signatures with the nonces reduced (nn_get_random_mod) and the messages hashed on the device; hash_type 0: msgs
        are digests of equal length"""
        n = len(privs) // self.qlen
        if hash_type:
            slots, stride = self.msg_slots(msgs, stride)
        else:
            slots, stride = b"".join(msgs), (len(msgs[0]) if msgs else 32)
        sigs, st = C.create_string_buffer(max(1, 2 * self.qlen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecdsa_sign_msg_batch(self.ctx.h, self.h, n, privs, nonce_raw, hash_type, slots, stride, sigs, st),
             "ec_ecdsa_sign_msg_batch")
        return sigs.raw[:2 * self.qlen * n], st.raw[:n]

    def key_pair_gen_raw(self, raw):
        """x = nn_get_random_mod value of the item's 2 * qlen random bytes, Y = [x]G: (privs, pubs affine, status)"""
        n = len(raw) // (2 * self.qlen)
        pr, pb, st = C.create_string_buffer(max(1, self.qlen * n)), C.create_string_buffer(max(1, 2 * self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_key_pair_gen_raw_batch(self.ctx.h, self.h, n, raw, pr, pb, st), "ec_key_pair_gen_raw_batch")
        return pr.raw[:self.qlen * n], pb.raw[:2 * self.clen * n], st.raw[:n]

    def ecccdh(self, privs, peers):
        n = len(privs) // self.qlen
        sec = C.create_string_buffer(max(1, self.clen * n))
        st = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_ecccdh_derive_batch(self.ctx.h, self.h, n, privs, peers, sec, st),
             "ec_ecccdh_derive_batch")
        return sec.raw[:self.clen * n], st.raw[:n]

    def xdh(self, k, u):
        """X25519 (WEI25519) / X448 (WEI448): little-endian scalars and u coordinates"""
        n = len(k) // self.clen
        out = C.create_string_buffer(max(1, self.clen * n))
        st = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_xdh_batch(self.ctx.h, self.h, n, k, u, out, st), "ec_xdh_batch")
        return out.raw[:self.clen * n], st.raw[:n]

    def eddsa_verify(self, pubkeys, sigs, hram, hram_len=None):
        """Ed25519 (WEI25519 handle): 32-byte keys, 64-byte signatures, hram = SHA-512(dom2 || R || A || PH(M));
        Ed448 (WEI448 handle): 57-byte keys, 114-byte signatures, hram = SHAKE256(dom4 || R || A || PH(M), 114)"""
        klen = 57 if self.clen == 56 else self.clen
        hram_len = hram_len or (114 if self.clen == 56 else 64)
        n = len(pubkeys) // klen
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_verify_batch(self.ctx.h, self.h, n, pubkeys, sigs, hram, hram_len, res),
             "ec_eddsa_verify_batch")
        return res.raw[:n]

    def eddsa_sign_R(self, r_hash):
        """EdDSA signing, step 1: n x 64-byte SHA-512(dom2 || prefix || PH(M)) -> (n x 32 encoded R, status); on the WEI448
        handle n x 114-byte SHAKE256(dom4 || prefix || PH(M)) -> n x 57"""
        hl, kl = (114, 57) if self.clen == 56 else (64, 32)
        n = len(r_hash) // hl
        out, st = C.create_string_buffer(max(1, kl * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_sign_R_batch(self.ctx.h, self.h, n, r_hash, out, st), "ec_eddsa_sign_R_batch")
        return out.raw[:kl * n], st.raw[:n]

    def eddsa_sign_S(self, r_hash, hram, a_scalars):
        """EdDSA signing, step 2: S = (r + hram * a) mod q, n x 32 (Ed25519) / n x 57 (Ed448) bytes little-endian"""
        hl, kl = (114, 57) if self.clen == 56 else (64, 32)
        n = len(r_hash) // hl
        out = C.create_string_buffer(max(1, kl * n))
        _chk(self.L, self.L.ec_eddsa_sign_S_batch(self.ctx.h, self.h, n, r_hash, hram, a_scalars, out), "ec_eddsa_sign_S_batch")
        return out.raw[:kl * n]

    def eddsa_encode_points(self, points_prj):
        """eddsa_export_pub_key in batch: projective Weierstrass X || Y || Z -> the 32 / 57-byte EdDSA encodings, status"""
        kl = 57 if self.clen == 56 else 32
        n = len(points_prj) // (3 * self.clen)
        out, st = C.create_string_buffer(max(1, kl * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_eddsa_encode_point_batch(self.ctx.h, self.h, n, points_prj, out, st), "ec_eddsa_encode_point_batch")
        return out.raw[:kl * n], st.raw[:n]

    def eddsa_verify_all(self, pubkeys, sigs, hram, hram_len=None):
        """ec_verify_batch's whole-batch predicate: (all_valid, index of the first rejected item or n)"""
        klen = 57 if self.clen == 56 else self.clen
        hram_len = hram_len or (114 if self.clen == 56 else 64)
        n = len(pubkeys) // klen
        ok, first = C.c_int(0), C.c_uint32(0)
        _chk(self.L, self.L.ec_eddsa_verify_all_batch(self.ctx.h, self.h, n, pubkeys, sigs, hram, hram_len,
                                                       C.byref(ok), C.byref(first)), "ec_eddsa_verify_all_batch")
        return bool(ok.value), first.value

    def debug_eddsa_msm(self, pubkeys, sigs, hram, seed):
        """test hook: (accept, z_i as n x 16 bytes little-endian, [X, Y, Z, T] of the sum before the cofactor as integers mod p)"""
        n = len(pubkeys) // 32
        acc = C.c_int(0)
        z = C.create_string_buffer(16 * n)
        w = (C.c_uint32 * 36)()
        _chk(self.L, self.L.ecamd_debug_eddsa_msm(self.ctx.h, self.h, n, pubkeys, sigs, hram, seed, C.byref(acc),
                                                   C.cast(z, C.c_void_p), C.cast(w, C.c_void_p)), "ecamd_debug_eddsa_msm")
        p = 2 ** 255 - 19
        coords = [sum(int(w[9 * c + i]) << (29 * i) for i in range(9)) % p for c in range(4)]
        return bool(acc.value), z.raw, coords

    def schnorr_verify_all(self, s, ne, keys_aff, r, r_fmt=0):
        """ec_schnorr_verify_all_batch: True iff sum z_i ([s_i]G + [ne_i]Y_i - R_i) vanishes (the BIP0340 / ECFSDSA batch equation as one
        multi-scalar multiplication); False = not decided here.  r_fmt 1: r holds x coordinates, R_i has the even y."""
        n = len(s) // self.qlen
        ok = C.c_int(0)
        _chk(self.L, self.L.ec_schnorr_verify_all_batch(self.ctx.h, self.h, n, s, ne, keys_aff, r, r_fmt, C.byref(ok)), "ec_schnorr_verify_all_batch")
        return bool(ok.value)

    def schnorr_msm_available(self, r_fmt=0):
        return bool(self.L.ec_schnorr_verify_all_available(self.h, r_fmt))

    def eddsa_verify_msg_prj_all(self, keys_prj, sigs, slots, stride, a_offset):
        """ec_eddsa_verify_msg_prj_all_batch: ec_verify_batch's one bit for plain Ed25519 from projective keys, signatures and hash inputs;
        True = the whole batch is valid, False = not decided here"""
        n = len(slots) // stride
        ok = C.c_int(0)
        _chk(self.L, self.L.ec_eddsa_verify_msg_prj_all_batch(self.ctx.h, self.h, n, keys_prj, sigs, slots, stride, a_offset, C.byref(ok)),
             "ec_eddsa_verify_msg_prj_all_batch")
        return bool(ok.value)

    def schnorr_verify_msg_all(self, keys, key_fmt, sigs, r_fmt, hash_type, slots, stride, x_offset):
        """ec_schnorr_verify_msg_all_batch: the BIP0340 / ECFSDSA batch from keys, signatures and hash inputs (hashes, q - e and the
        key's lift_x on the device); True = the whole batch is valid, False = not decided here"""
        n = len(slots) // stride
        ok = C.c_int(0)
        _chk(self.L, self.L.ec_schnorr_verify_msg_all_batch(self.ctx.h, self.h, n, keys, key_fmt, sigs, r_fmt, hash_type, slots, stride,
                                                             x_offset, C.byref(ok)), "ec_schnorr_verify_msg_all_batch")
        return bool(ok.value)

    def debug_schnorr_msm(self, s, ne, keys_aff, r, r_fmt, seed):
        """test hook: (accept, z_i as n x 16 bytes little-endian, the "sum is infinity" word of the lanes' sum)"""
        n = len(s) // self.qlen
        acc = C.c_int(0)
        z = C.create_string_buffer(16 * n)
        nw = self.L.ecamd_debug_schnorr_msm_words(self.h)
        w = (C.c_uint32 * max(1, nw))()
        _chk(self.L, self.L.ecamd_debug_schnorr_msm(self.ctx.h, self.h, n, s, ne, keys_aff, r, r_fmt, seed, C.byref(acc),
                                                     C.cast(z, C.c_void_p), C.cast(w, C.c_void_p)), "ecamd_debug_schnorr_msm")
        return bool(acc.value), z.raw, int(w[nw - 4]) if nw else None

    # -- device-pointer forms (torch tensors' data_ptr()); see include/libecc_amd.h for which ones synchronise --
    def eddsa_verify_all_dev(self, n, d_pubs, d_sigs, d_hram, d_verdict, stream=None):
        _chk(self.L, self.L.ec_eddsa_verify_all_batch_dev(self.ctx.h, self.h, n, d_pubs, d_sigs, d_hram, 114 if self.clen == 56 else 64, d_verdict, stream),
             "ec_eddsa_verify_all_batch_dev")

    def schnorr_verify_all_dev(self, n, d_s, d_ne, d_keys, d_r, r_fmt, d_verdict, stream=None):
        _chk(self.L, self.L.ec_schnorr_verify_all_batch_dev(self.ctx.h, self.h, n, d_s, d_ne, d_keys, d_r, r_fmt, d_verdict, stream),
             "ec_schnorr_verify_all_batch_dev")

    def ecdsa_verify_dev(self, n, d_pubs, d_sigs, d_digests, hlen, d_result, stream=None):
        _chk(self.L, self.L.ec_ecdsa_verify_batch_dev(self.ctx.h, self.h, n, d_pubs, d_sigs, d_digests, hlen, d_result,
                                                       stream), "ec_ecdsa_verify_batch_dev")

    def eddsa_verify_dev(self, n, d_pubs, d_sigs, d_hram, d_result, stream=None, hram_len=64):
        _chk(self.L, self.L.ec_eddsa_verify_batch_dev(self.ctx.h, self.h, n, d_pubs, d_sigs, d_hram, hram_len, d_result,
                                                       stream), "ec_eddsa_verify_batch_dev")

    def ecdsa_sign_dev(self, n, d_privs, d_nonces, d_digests, hlen, d_sigs, d_status, stream=None):
        _chk(self.L, self.L.ec_ecdsa_sign_batch_dev(self.ctx.h, self.h, n, d_privs, d_nonces, d_digests, hlen, d_sigs,
                                                     d_status, stream), "ec_ecdsa_sign_batch_dev")

    def ecccdh_dev(self, n, d_privs, d_peers, d_secrets, d_status, stream=None):
        _chk(self.L, self.L.ec_ecccdh_derive_batch_dev(self.ctx.h, self.h, n, d_privs, d_peers, d_secrets, d_status,
                                                        stream), "ec_ecccdh_derive_batch_dev")

    def xdh_dev(self, n, d_k, d_u, d_out, d_status, stream=None):
        _chk(self.L, self.L.ec_xdh_batch_dev(self.ctx.h, self.h, n, d_k, d_u, d_out, d_status, stream),
             "ec_xdh_batch_dev")

    # -- point wire formats: 0 affine X || Y, 1 projective X || Y || Z --
    def scalar_mult_fmt(self, scalars, points, in_fmt, out_fmt, slen=None):
        slen = slen or self.qlen
        n = len(scalars) // slen
        w = (3 if out_fmt else 2) * self.clen
        out, st = C.create_string_buffer(max(1, w * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_mul_batch_fmt(self.ctx.h, self.h, n, scalars, slen, points, in_fmt, out, out_fmt, st),
             "ec_prj_pt_mul_batch_fmt")
        return out.raw[:w * n], st.raw[:n]

    def unique(self, points, in_fmt, out_fmt):
        n = len(points) // ((3 if in_fmt else 2) * self.clen)
        w = (3 if out_fmt else 2) * self.clen
        out, st = C.create_string_buffer(max(1, w * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_unique_batch(self.ctx.h, self.h, n, points, in_fmt, out, out_fmt, st),
             "ec_prj_pt_unique_batch")
        return out.raw[:w * n], st.raw[:n]

    def pt_op_fmt(self, op, p1, p2, in_fmt, out_fmt):
        """prj_pt_add (op 0) / prj_pt_dbl (1) / prj_pt_is_on_curve (2) / prj_pt_neg (3) in either wire format, prj_pt_cmp (4) /
        prj_pt_eq_or_opp (5) with one predicate byte per item: (out, status)"""
        iw, ow = (3 if in_fmt else 2) * self.clen, (1 if op >= 4 else (3 if out_fmt else 2) * self.clen)
        n = len(p1) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_op_batch_fmt(self.ctx.h, self.h, op, n, p1, p2, in_fmt, None if op == 2 else out, out_fmt, st),
             "ec_prj_pt_op_batch_fmt")
        return (b"" if op == 2 else out.raw[:ow * n]), st.raw[:n]

    def unprotected_mult(self, scalars, slen, points, in_fmt, out_fmt, broadcast=False):
        """_prj_pt_unprotected_mult per item (broadcast: one scalar of slen bytes for every point, check_prj_pt_order's use)"""
        iw, ow = (3 if in_fmt else 2) * self.clen, (3 if out_fmt else 2) * self.clen
        n = len(points) // iw
        out, st = C.create_string_buffer(max(1, ow * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_prj_pt_unprotected_mult_batch(self.ctx.h, self.h, n, scalars, slen, 0 if broadcast else slen, points, in_fmt,
                                                             out, out_fmt, st), "ec_prj_pt_unprotected_mult_batch")
        return out.raw[:ow * n], st.raw[:n]

    def y_from_x(self, xs):
        """aff_pt_y_from_x: (y1, y2, status), y1 the root the reference's fp_sqrt returns first"""
        n = len(xs) // self.clen
        y1, y2 = C.create_string_buffer(max(1, self.clen * n)), C.create_string_buffer(max(1, self.clen * n))
        st = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_aff_pt_y_from_x_batch(self.ctx.h, self.h, n, xs, y1, y2, st), "ec_aff_pt_y_from_x_batch")
        return y1.raw[:self.clen * n], y2.raw[:self.clen * n], st.raw[:n]

    def decompress(self, comp):
        """SEC 1 compressed points (0x02 / 0x03 || x) -> affine X || Y + status"""
        n = len(comp) // (self.clen + 1)
        out, st = C.create_string_buffer(max(1, 2 * self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_point_decompress_batch(self.ctx.h, self.h, n, comp, out, st), "ec_point_decompress_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]

    def structured_sigs(self, sigs, slen, alg_type, hash_type):
        n = len(sigs) // slen
        out, st = C.create_string_buffer(max(1, (slen - 3) * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_structured_sig_import_batch(self.h, n, sigs, slen, alg_type, hash_type, out, st),
             "ec_structured_sig_import_batch")
        return out.raw[:(slen - 3) * n], st.raw[:n]

    def structured_key_pairs(self, keys, klen, alg_type):
        """structured private keys -> (private scalars n x qlen, public keys X || Y || Z, status)"""
        n = len(keys) // klen
        pv, pb, st = (C.create_string_buffer(max(1, self.qlen * n)), C.create_string_buffer(max(1, 3 * self.clen * n)),
                      C.create_string_buffer(max(1, n)))
        _chk(self.L, self.L.ec_structured_key_pair_import_batch(self.ctx.h, self.h, n, keys, klen, alg_type, pv, pb, st),
             "ec_structured_key_pair_import_batch")
        return pv.raw[:self.qlen * n], pb.raw[:3 * self.clen * n], st.raw[:n]

    def structured_pub_keys(self, keys, alg_type):
        """libecc's structured public keys (3 header bytes + X || Y || Z) -> affine X || Y + status"""
        klen = 3 + 3 * self.clen
        n = len(keys) // klen
        out, st = C.create_string_buffer(max(1, 2 * self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ec_structured_pub_key_import_batch(self.ctx.h, self.h, n, keys, klen, alg_type, out, st),
             "ec_structured_pub_key_import_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]



class Multi:
    """ecamd_multi: one context and one host thread per listed device; batches are cut into contiguous shards.
    A device may be listed several times (its shards then share that GPU)."""

    def __init__(self, devices=None):
        self.L = load_library()
        self.h = C.c_void_p()
        if devices is None:
            rc = self.L.ecamd_multi_create(C.byref(self.h), None, 0)
        else:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.L.ecamd_multi_create(C.byref(self.h), arr, len(devices))
        _chk(self.L, rc, "ecamd_multi_create")
        self.size = self.L.ecamd_multi_size(self.h)

    def close(self):
        if self.h:
            self.L.ecamd_multi_destroy(self.h)
            self.h = C.c_void_p()

    def shard_range(self, n, rank):
        lo, hi = C.c_uint32(0), C.c_uint32(0)
        self.L.ecamd_multi_shard_range(n, rank, self.size, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def curve(self, name):
        return MultiCurve(self, name)


class MultiCurve:
    def __init__(self, multi, name):
        self.m, self.L = multi, multi.L
        self.h = C.c_void_p()
        _chk(self.L, self.L.ecamd_multi_curve_by_name(multi.h, name.encode(), C.byref(self.h)), "ecamd_multi_curve_by_name")
        self.clen = self.L.ecamd_multi_curve_coord_len(self.h)
        self.qlen = self.L.ecamd_multi_curve_order_len(self.h)

    def free(self):
        if self.h:
            self.L.ecamd_multi_curve_free(self.h)
            self.h = C.c_void_p()

    def scalar_mult(self, scalars, points=None, slen=None):
        slen = slen or self.qlen
        n = len(scalars) // slen
        out, st = C.create_string_buffer(max(1, 2 * self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_prj_pt_mul_batch(self.m.h, self.h, n, scalars, slen, points, out, st),
             "ecamd_multi_prj_pt_mul_batch")
        return out.raw[:2 * self.clen * n], st.raw[:n]

    def ecdsa_verify(self, pubs, sigs, digests, hlen):
        n = len(pubs) // (2 * self.clen)
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_ecdsa_verify_batch(self.m.h, self.h, n, pubs, sigs, digests, hlen, res),
             "ecamd_multi_ecdsa_verify_batch")
        return res.raw[:n]

    def ecdsa_sign(self, privs, nonces, digests, hlen):
        n = len(privs) // self.qlen
        sigs, st = C.create_string_buffer(max(1, 2 * self.qlen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_ecdsa_sign_batch(self.m.h, self.h, n, privs, nonces, digests, hlen, sigs, st),
             "ecamd_multi_ecdsa_sign_batch")
        return sigs.raw[:2 * self.qlen * n], st.raw[:n]

    def ecccdh(self, privs, peers):
        n = len(privs) // self.qlen
        sec, st = C.create_string_buffer(max(1, self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_ecccdh_derive_batch(self.m.h, self.h, n, privs, peers, sec, st),
             "ecamd_multi_ecccdh_derive_batch")
        return sec.raw[:self.clen * n], st.raw[:n]

    def xdh(self, k, u):
        n = len(k) // self.clen
        out, st = C.create_string_buffer(max(1, self.clen * n)), C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_xdh_batch(self.m.h, self.h, n, k, u, out, st), "ecamd_multi_xdh_batch")
        return out.raw[:self.clen * n], st.raw[:n]

    def eddsa_verify(self, pubkeys, sigs, hram, hram_len=None):
        klen = 57 if self.clen == 56 else self.clen
        hram_len = hram_len or (114 if self.clen == 56 else 64)
        n = len(pubkeys) // klen
        res = C.create_string_buffer(max(1, n))
        _chk(self.L, self.L.ecamd_multi_eddsa_verify_batch(self.m.h, self.h, n, pubkeys, sigs, hram, hram_len, res),
             "ecamd_multi_eddsa_verify_batch")
        return res.raw[:n]

    def schnorr_verify_all(self, s, ne, keys_aff, r, r_fmt=0):
        n = len(s) // self.qlen
        ok = C.c_int(0)
        _chk(self.L, self.L.ecamd_multi_schnorr_verify_all_batch(self.m.h, self.h, n, s, ne, keys_aff, r, r_fmt, C.byref(ok)),
             "ecamd_multi_schnorr_verify_all_batch")
        return bool(ok.value)

    def eddsa_verify_all(self, pubkeys, sigs, hram, hram_len=None):
        klen = 57 if self.clen == 56 else self.clen
        hram_len = hram_len or (114 if self.clen == 56 else 64)
        n = len(pubkeys) // klen
        ok, first = C.c_int(0), C.c_uint32(0)
        _chk(self.L, self.L.ecamd_multi_eddsa_verify_all_batch(self.m.h, self.h, n, pubkeys, sigs, hram, hram_len,
                                                                C.byref(ok), C.byref(first)), "ecamd_multi_eddsa_verify_all_batch")
        return bool(ok.value), first.value
