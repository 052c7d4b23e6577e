"""ctypes prototypes for include/lasso_hip.h (shared by the real device library and, in tests, the oracle's mock)."""
import ctypes as C

u64 = C.c_uint64
u32 = C.c_uint32
i32 = C.c_int32
sz = C.c_size_t
vp = C.c_void_p


class Strategy(C.Structure):
    _fields_ = [("kind", i32), ("c", u32), ("log_m", u32), ("log_r", u32)]


KINDS = {"and": 0, "or": 1, "xor": 2, "lt": 3, "range": 4, "spark": 5}   # "spark" = LASSO_SPARK_UNCONFIRMED (include/lasso_hip.h): not in the reference snapshot
K_BIND, K_CUBIC, K_COMBINE, K_EQ, K_GP, K_FINGERPRINT, K_DOT, K_MATVEC, K_MSM, K_MISC, K_MSM_DIRECT, K_COUNT = range(12)
KERNEL_NAMES = ["bind_top(+fused linear round)", "sumcheck_cubic_round(+fused bind)", "sumcheck_combine", "eq_evals", "gp_build", "fingerprint", "multi_dot", "matvec_left", "msm_commit(bucket)", "misc", "msm_opening(direct)"]


def declare(lib):
    P = C.POINTER
    sig = {
        "lasso_ctx_create": (i32, [i32, P(vp)]),
        "lasso_ctx_create_background": (i32, [i32, i32, P(vp)]),
        "lasso_ctx_destroy": (None, [vp]),
        "lasso_last_error": (C.c_char_p, [vp]),
        "lasso_alloc": (i32, [vp, sz, P(vp)]),
        "lasso_free": (i32, [vp, vp]),
        "lasso_upload": (i32, [vp, vp, vp, sz]),
        "lasso_download": (i32, [vp, vp, vp, sz]),
        "lasso_copy": (i32, [vp, vp, vp, sz]),
        "lasso_zero": (i32, [vp, vp, sz]),
        "lasso_sync": (i32, [vp]),
        "lasso_abort": (i32, [vp]),
        "lasso_stream": (vp, [vp]),
        "lasso_prof_enable": (i32, [vp, i32]),
        "lasso_prof_reset": (i32, [vp]),
        "lasso_prof_get": (i32, [vp, i32, P(u64), P(C.c_double), P(C.c_double)]),
        "lasso_prof_get_large": (i32, [vp, i32, P(u64), P(C.c_double), P(C.c_double)]),
        "lasso_prof_get_units": (i32, [vp, i32, i32, P(C.c_double)]),
        "lasso_wait_stats": (i32, [vp, P(u64), P(C.c_double), i32]),
        "lasso_mem_stats": (i32, [vp, P(u64), P(u64), i32]),
        "lasso_trim": (i32, [vp]),
        "lasso_fr_from_u32": (i32, [vp, vp, sz, vp]),
        "lasso_fr_to_u32": (i32, [vp, vp, sz, vp, vp]),
        "lasso_gather": (i32, [vp, vp, vp, sz, vp]),
        "lasso_eq_evals": (i32, [vp, vp, u32, vp]),
        "lasso_eq_evals_scaled": (i32, [vp, vp, u32, vp, vp]),
        "lasso_bind_top": (i32, [vp, P(vp), u32, sz, vp]),
        "lasso_sumcheck_cubic_round": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_cubic_eqw_round": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_cubic_eqw_round_fused": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp, vp]),
        "lasso_sumcheck_cubic_eqw2_begin": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp]),
        "lasso_result_wait": (i32, [vp, vp, sz]),
        "lasso_defer_next": (i32, [vp]),
        "lasso_sumcheck_cubic_tail_begin": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_cubic_tail_next": (i32, [vp, vp]),
        "lasso_sumcheck_linear_tail_begin": (i32, [vp, P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_combine_round": (i32, [vp, P(Strategy), P(vp), vp, sz, u32, vp]),
        "lasso_sumcheck_combine_round_lt_scaled": (i32, [vp, P(Strategy), P(vp), vp, sz, u32, vp]),
        "lasso_lt_prescale": (i32, [vp, P(Strategy), P(vp), P(vp), sz]),
        "lasso_sumcheck_combine_round_lt_u32": (i32, [vp, P(Strategy), P(vp), vp, sz, u32, vp]),
        "lasso_sumcheck_linear_eqw_round": (i32, [vp, P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_linear_eqw_round_fused": (i32, [vp, P(vp), u32, vp, sz, vp, vp]),
        "lasso_sumcheck_linear_eqw_round_fused_from": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp, vp]),
        "lasso_sumcheck_linear_eqw_round_u32": (i32, [vp, P(vp), u32, vp, sz, vp]),
        "lasso_sumcheck_linear_eqw_round_fused_from_u32": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp, vp]),
        "lasso_combine_claim": (i32, [vp, P(Strategy), P(vp), vp, sz, vp]),
        "lasso_multi_dot": (i32, [vp, P(vp), u32, vp, sz, vp]),
        "lasso_read_heads": (i32, [vp, P(vp), u32, vp]),
        "lasso_read_runs": (i32, [vp, P(vp), u32, u32, vp]),
        "lasso_tail_handover_next": (i32, [vp, u32]),
        "lasso_rounds_ahead_ok": (i32, [vp]),
        "lasso_layer_ahead_ok": (i32, [vp]),
        "lasso_sumcheck_cubic_eqw2_begin_eq_ahead": (i32, [vp, P(vp), P(vp), u32, vp, sz, u32]),
        "lasso_sumcheck_cubic_tail_begin_eq_ahead": (i32, [vp, P(vp), P(vp), u32, sz, u32]),
        "lasso_point_post": (i32, [vp, vp, u32, vp]),
        "lasso_point_cancel": (i32, [vp]),
        "lasso_sumcheck_cubic_eqw2_begin_ahead": (i32, [vp, P(vp), P(vp), u32, vp, sz]),
        "lasso_challenge_post": (i32, [vp, vp]),
        "lasso_sumcheck_linear_eqw_round_fused_ahead": (i32, [vp, P(vp), u32, vp, sz]),
        "lasso_sumcheck_cubic_tail_begin_ahead": (i32, [vp, P(vp), P(vp), u32, vp, sz]),
        "lasso_gp_build": (i32, [vp, vp, sz]),
        "lasso_fingerprint_ops": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_fingerprint_ops_gp": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_fingerprint_ops_gp_upper": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_fingerprint_ops_strips": (i32, [vp, vp, vp, vp, sz, vp, vp, u32, sz, sz, vp, vp]),
        "lasso_fingerprint_ops_gp_upper_u32": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_fingerprint_ops_strips_u32": (i32, [vp, vp, vp, vp, sz, vp, vp, u32, sz, sz, vp, vp]),
        "lasso_fingerprint_mem": (i32, [vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_fingerprint_mem_slab": (i32, [vp, vp, vp, sz, u32, u32, vp, vp, vp, vp]),
        "lasso_densify_dim_slab": (i32, [vp, vp, sz, sz, sz, sz, u32, u32, u32, vp, vp, vp, vp]),
        "lasso_matvec_left_dev": (i32, [vp, vp, vp, sz, sz, vp]),
        "lasso_fr_to_bytes": (i32, [vp, vp, sz, vp]),
        "lasso_msm_dev_scaled": (i32, [vp, vp, vp, sz, vp, vp, vp]),
        "lasso_densify_dim": (i32, [vp, vp, sz, sz, sz, sz, u32, vp, vp, vp, vp]),
        "lasso_matvec_left": (i32, [vp, vp, vp, sz, sz, vp]),
        "lasso_bases_create": (i32, [vp, vp, sz, P(vp)]),
        "lasso_bases_create_opt": (i32, [vp, vp, sz, i32, P(vp)]),
        "lasso_bases_destroy": (None, [vp, vp]),
        "lasso_hyrax_commit": (i32, [vp, vp, sz, sz, vp, vp]),
        "lasso_hyrax_commit_compressed": (i32, [vp, vp, sz, sz, vp, vp]),
        "lasso_hyrax_commit_compressed_u32": (i32, [vp, vp, u32, sz, sz, vp, vp]),
        "lasso_bases_has_direct": (i32, [vp]),
        "lasso_bullet_round_slab": (i32, [vp, vp, sz, u32, u32, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_msm_dev_slab": (i32, [vp, vp, vp, sz, u32, u32, vp, vp, vp]),
        "lasso_sumcheck_tail_capacity": (u32, []),
        "lasso_sumcheck_cubic_eqw2_begin_eq": (i32, [vp, P(vp), P(vp), u32, vp, sz, vp, u32, vp]),
        "lasso_sumcheck_cubic_tail_begin_eq": (i32, [vp, P(vp), P(vp), u32, sz, vp, u32, vp]),
        "lasso_rccl_available": (i32, []),
        "lasso_rccl_unique_id": (i32, [vp]),
        "lasso_rccl_init": (i32, [vp, i32, i32, vp]),
        "lasso_rccl_ready": (i32, [vp]),
        "lasso_rccl_shutdown": (i32, [vp]),
        "lasso_rccl_selftest": (i32, [vp]),
        "lasso_ctx_device_uuid": (i32, [vp, vp]),
        "lasso_bullet_tail_ahead_ok": (i32, [vp, vp]),
        "lasso_bullet_tail_ahead": (i32, [vp, vp, sz, vp, vp, vp, sz, vp, vp, vp]),
        "lasso_bases_prepare": (i32, [vp, vp, u32]),
        "lasso_rccl_allgather": (i32, [vp, vp, vp, sz]),
        "lasso_point_row_bytes": (sz, []),
        "lasso_hyrax_commit_rows_dev": (i32, [vp, vp, sz, sz, vp, vp]),
        "lasso_points_reduce_compress": (i32, [vp, vp, u32, sz, vp]),
        "lasso_gather_u32": (i32, [vp, vp, vp, sz, vp]),
        "lasso_materialize_subtable_u32": (i32, [vp, P(Strategy), u32, vp]),
        "lasso_msm": (i32, [vp, vp, vp, sz, vp]),
        "lasso_msm_dev": (i32, [vp, vp, vp, sz, vp]),
        "lasso_inner_products_lr": (i32, [vp, vp, vp, sz, vp]),
        "lasso_bullet_lr": (i32, [vp, vp, sz, vp, sz, vp, vp, vp]),
        "lasso_bullet_round": (i32, [vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]),
        "lasso_bullet_ahead_ok": (i32, [vp, vp]),
        "lasso_bullet_round_ahead": (i32, [vp, vp, sz, vp, vp, vp, vp, vp, vp, sz, vp]),
        "lasso_bullet_post": (i32, [vp, vp, vp]),
        "lasso_bullet_fold": (i32, [vp, vp, vp, sz, vp, sz, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError here = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return sorted(sig)


HEADER_SYMBOLS = None
