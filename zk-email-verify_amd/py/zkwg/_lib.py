"""ctypes binding of libzkwg.so (the C-ABI declared in include/zkwg.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be
loaded this module raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libzkwg.so"))


class RegexSource(C.Structure):   # zkwg_regex_source
    _fields_ = [("circom_path", C.c_char_p), ("include_dirs", C.c_char_p), ("template_name", C.c_char_p)]


class Config(C.Structure):
    _fields_ = [
        ("main_kind", C.c_uint32),
        ("max_header", C.c_uint32),
        ("max_body", C.c_uint32),
        ("n", C.c_uint32),
        ("k", C.c_uint32),
        ("ignore_body_hash_check", C.c_uint32),
        ("enable_header_masking", C.c_uint32),
        ("enable_body_masking", C.c_uint32),
        ("remove_soft_line_breaks", C.c_uint32),
        ("layout", C.c_uint32),
    ]


class DkimBatch(C.Structure):
    _fields_ = [("headers", C.c_void_p), ("header_len", C.c_void_p), ("bodies", C.c_void_p), ("body_len", C.c_void_p),
                ("body_hash_b64", C.c_void_p), ("pubkey_be", C.c_void_p), ("signature_be", C.c_void_p),
                ("selector", C.c_void_p), ("header_stride", C.c_uint32), ("body_stride", C.c_uint32),
                ("selector_len", C.c_uint32)]


MAIN_EMAIL_VERIFIER, MAIN_SHA256_BYTES, MAIN_RSA_VERIFIER, MAIN_FP_MUL = 0, 1, 2, 3
(IN_HEADER, IN_BODY, IN_PRECOMPUTED_SHA, IN_PUBKEY, IN_SIGNATURE, IN_MESSAGE,
 IN_HEADER_LEN, IN_BODY_LEN, IN_BODY_HASH_INDEX, IN_HEADER_MASK, IN_BODY_MASK, IN_DECODED_BODY,
 IN_RANGE_FLAGS) = range(13)

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"zkwg: HIP library not built: {LIB_PATH} missing "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C zk-email-verify_amd/csrc`)")
    # PyTorch (when present) bundles its own HIP runtime; load it first so that libzkwg.so binds to
    # the same libamdhip64 instance -- two HIP runtimes in one process cannot both open the GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    sig = {
        "zkwg_abi_version": (i32, []),
        "zkwg_strerror": (C.c_char_p, [i32]),
        "zkwg_circuit_create": (i32, [C.POINTER(Config), i32, C.POINTER(vp)]),
        "zkwg_circuit_create_sym": (i32, [C.POINTER(Config), i32, C.c_char_p, u64, C.c_char_p, u64, C.POINTER(vp)]),
        "zkwg_circuit_create_full": (i32, [C.POINTER(Config), i32, C.c_char_p, u64, C.c_char_p, u64, vp, u64, C.POINTER(vp)]),
        "zkwg_circuit_create_regex": (i32, [C.POINTER(Config), i32, C.POINTER(RegexSource), C.c_char_p, u64, C.c_char_p, u64, vp, u64, C.POINTER(vp)]),
        "zkwg_regex_info": (i32, [vp, vp]),
        "zkwg_linear_rows": (u64, [vp]),
        "zkwg_layout_map": (u64, [vp, vp, u64]),
        "zkwg_image_layout": (i32, [vp, u64, vp]),
        "zkwg_segment_table": (u64, [vp, vp, u64]),
        "zkwg_inverse_table_half": (u32, [vp]),
        "zkwg_linear_complete_host": (i32, [vp, vp]),
        "zkwg_o0_gather_host": (i32, [vp, vp, vp]),
        "zkwg_last_error": (C.c_char_p, []),
        "zkwg_circuit_destroy": (None, [vp]),
        "zkwg_witness_len": (u64, [vp]),
        "zkwg_witness_bytes": (u64, [vp]),
        "zkwg_num_public": (u32, [vp]),
        "zkwg_input_stride": (u64, [vp]),
        "zkwg_input_offset": (u64, [vp, i32]),
        "zkwg_scratch_bytes": (u64, [vp, u64]),
        "zkwg_scratch_bytes_standard": (u64, [vp, u64]),
        "zkwg_pack_input": (i32, [vp, vp, vp, u32, vp, u32, vp, vp, vp, vp, u32]),
        "zkwg_pack_field": (i32, [vp, vp, i32, u64, vp, u64]),
        "zkwg_pack_masks": (i32, [vp, vp, vp, vp]),
        "zkwg_pack_decoded_body": (i32, [vp, vp, vp]),
        "zkwg_calculate_batch": (i32, [vp, vp, u64, vp, u64, vp, u64]),
        "zkwg_generate_inputs_device": (i32, [vp, vp, u64, vp, vp, vp]),
        "zkwg_alloc_pinned": (vp, [u64]),
        "zkwg_stream_create_masked": (vp, [i32, C.POINTER(C.c_uint32), i32]),
        "zkwg_stream_destroy": (None, [vp]),
        "zkwg_expand_host": (i32, [vp, vp, u64, vp, u64, u64, vp, u64, i32]),
        "zkwg_set_host_expand": (i32, [vp, i32]),
        "zkwg_free_pinned": (None, [vp]),
        "zkwg_calculate_batch_device": (i32, [vp, vp, u64, vp, u64, vp, vp, vp]),
        "zkwg_prepare_device": (i32, [vp, vp, u64, vp, vp, vp]),
        "zkwg_expand_device": (i32, [vp, vp, u64, vp, u64, u64, vp, u64, vp]),
        "zkwg_expand_montgomery_device": (i32, [vp, vp, u64, vp, u64, u64, vp, u64, vp]),
        "zkwg_circuit_attach_r1cs": (i32, [vp, C.c_char_p, u64]),
        "zkwg_abc_bytes": (u64, [vp]),
        "zkwg_expand_abc_device": (i32, [vp, vp, u64, vp, u64, u64, i32, vp, u64, vp]),
        "zkwg_expand_abc_host": (i32, [vp, C.c_char_p, u64, vp, u64, u64, i32, vp, u64]),
        "zkwg_expand_full_host": (i32, [vp, C.c_char_p, u64, vp, u64, u64, vp, u64]),
        "zkwg_set_prepare_throttle": (i32, [vp, i32]),
        "zkwg_set_prepare_mask": (i32, [vp, C.c_uint32]),
        "zkwg_set_timing": (i32, [vp, i32]),
        "zkwg_last_kernel_ms": (i32, [vp, i32, C.POINTER(C.c_float)]),
        "zkwg_timing_summary": (i32, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
        "zkwg_num_kernels": (i32, [vp]),
        "zkwg_kernel_name": (C.c_char_p, [vp, i32]),
        "zkwg_kernel_slots": (u64, [vp, i32]),
        "zkwg_wtns_size": (u64, [vp]),
        "zkwg_write_wtns": (i32, [vp, vp, vp, u64]),
        "zkwg_write_sym": (u64, [vp, vp, u64]),
        "zkwg_r1cs_load": (i32, [vp, u64, i32, C.POINTER(vp)]),
        "zkwg_r1cs_destroy": (None, [vp]),
        "zkwg_r1cs_info": (i32, [vp, C.POINTER(C.c_uint64)]),
        "zkwg_r1cs_evaluate_device": (i32, [vp, vp, u64, u64, i32, vp, u64, vp]),
        "zkwg_check_constraints_device": (i32, [vp, vp, u64, u64, vp, vp]),
        "zkwg_check_constraints": (i32, [vp, vp, u64, u64, C.POINTER(C.c_uint64)]),
        "zkwg_convert_montgomery_device": (i32, [vp, u64, i32, vp]),
        "zkwg_shard_range": (None, [u64, i32, i32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "zkwg_multi_create": (i32, [C.POINTER(Config), C.POINTER(C.c_int), i32, C.POINTER(vp)]),
        "zkwg_multi_destroy": (None, [vp]),
        "zkwg_multi_devices": (i32, [vp]),
        "zkwg_multi_circuit": (vp, [vp, i32]),
        "zkwg_calculate_batch_multi": (i32, [vp, vp, u64, vp, u64, vp, vp, u64]),
        "zkwg_ntt_create": (i32, [i32, C.c_uint32, C.POINTER(vp)]),
        "zkwg_ntt_destroy": (None, [vp]),
        "zkwg_ntt_domain": (u64, [vp]),
        "zkwg_ntt_work_bytes": (u64, [vp, u64]),
        "zkwg_ntt_transform_device": (i32, [vp, vp, u64, i32, vp]),
        "zkwg_h_evaluations_device": (i32, [vp, vp, u64, u64, u64, vp, vp, u64, vp]),
        "zkwg_msm_create": (i32, [i32, vp, u64, i32, C.POINTER(vp)]),
        "zkwg_msm_destroy": (None, [vp]),
        "zkwg_msm_work_bytes": (u64, [vp]),
        "zkwg_msm_window_bits": (i32, [vp]),
        "zkwg_msm_g1_device": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "zkwg_msm_g2_device": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "zkwg_msm_create_g2": (i32, [i32, vp, u64, i32, C.POINTER(vp)]),
        "zkwg_msm_create_device": (i32, [i32, i32, vp, u64, i32, C.POINTER(vp)]),
        "zkwg_msm_group": (i32, [vp]),
        "zkwg_prover_create": (i32, [vp, i32, vp, u64, u64, vp, u32, C.POINTER(vp)]),
        "zkwg_prover_destroy": (None, [vp]),
        "zkwg_prover_prove_prepared": (i32, [vp, vp, u64, vp, vp, u64, vp, vp]),
        "zkwg_prover_prove_batch": (i32, [vp, vp, u64, vp, vp, vp]),
        "zkwg_device_alloc_chunked": (i32, [i32, u64, u64, C.POINTER(vp)]),
        "zkwg_device_free_chunked": (i32, [vp]),
        "zkwg_device_alloc_chunked_ex": (i32, [i32, u64, u64, u32, C.POINTER(vp), vp, u32, C.POINTER(u32)]),
        "zkwg_msm_enqueue_device": (i32, [vp, vp, i32, i32, vp, vp, vp]),
        "zkwg_msm_finish_host": (i32, [i32, vp, u64, vp]),
        "zkwg_msm_create_ex": (i32, [i32, i32, vp, i32, u64, i32, i32, u64, C.POINTER(vp)]),
        "zkwg_msm_work_bytes_batch": (u64, [vp, u64]),
        "zkwg_msm_lists_bytes": (u64, [vp, u64]),
        "zkwg_msm_estimate_work_bytes": (u64, [i32, u64, i32, i32]),
        "zkwg_msm_table_bytes": (u64, [vp]),
        "zkwg_msm_precomputed": (i32, [vp]),
        "zkwg_msm_enqueue_batch_device": (i32, [vp, vp, u64, u64, i32, i32, vp, vp, vp]),
        "zkwg_msm_classify_device": (i32, [vp, vp, u32, vp, u64, u64, u64, i32, i32, vp, vp]),
        "zkwg_msm_enqueue_lists_device": (i32, [vp, vp, u64, u64, i32, vp, i32, vp, vp, vp]),
        "zkwg_prover_create_zkey": (i32, [vp, i32, vp, u64, u32, C.POINTER(vp)]),
        "zkwg_prover_emails_per_series": (u32, [vp]),
        "zkwg_prover_contexts": (u32, [vp]),
        "zkwg_fixed_base_device": (i32, [i32, i32, vp, u64, vp, vp]),
        "zkwg_groth16_assemble": (i32, [vp] * 15),
        "zkwg_calculate_batch_resident": (i32, [vp, vp, u64, vp, vp, u64, u64, vp, vp]),
        "zkwg_resident_placement": (i32, [vp, C.POINTER(C.c_float), i32, C.POINTER(C.c_int)]),
        "zkwg_resident_release": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTS = [
    "zkwg_abi_version", "zkwg_strerror", "zkwg_circuit_create", "zkwg_circuit_create_sym", "zkwg_circuit_create_full", "zkwg_circuit_create_regex", "zkwg_regex_info", "zkwg_linear_rows", "zkwg_layout_map", "zkwg_image_layout", "zkwg_segment_table", "zkwg_inverse_table_half", "zkwg_linear_complete_host", "zkwg_o0_gather_host", "zkwg_last_error", "zkwg_circuit_destroy",
    "zkwg_witness_len", "zkwg_witness_bytes", "zkwg_num_public", "zkwg_input_stride",
    "zkwg_input_offset", "zkwg_scratch_bytes", "zkwg_scratch_bytes_standard", "zkwg_pack_input", "zkwg_pack_field", "zkwg_pack_masks", "zkwg_pack_decoded_body", "zkwg_calculate_batch",
    "zkwg_generate_inputs_device", "zkwg_stream_create_masked", "zkwg_stream_destroy", "zkwg_expand_host", "zkwg_set_host_expand", "zkwg_alloc_pinned", "zkwg_free_pinned", "zkwg_calculate_batch_device", "zkwg_prepare_device", "zkwg_expand_device", "zkwg_expand_montgomery_device", "zkwg_circuit_attach_r1cs", "zkwg_abc_bytes", "zkwg_expand_abc_device", "zkwg_expand_abc_host", "zkwg_expand_full_host", "zkwg_set_prepare_throttle", "zkwg_set_prepare_mask", "zkwg_set_timing", "zkwg_last_kernel_ms", "zkwg_timing_summary", "zkwg_num_kernels",
    "zkwg_kernel_name", "zkwg_kernel_slots", "zkwg_wtns_size", "zkwg_write_wtns", "zkwg_write_sym",
    "zkwg_r1cs_load", "zkwg_r1cs_destroy", "zkwg_r1cs_info", "zkwg_check_constraints_device", "zkwg_r1cs_evaluate_device", "zkwg_check_constraints",
    "zkwg_convert_montgomery_device", "zkwg_shard_range", "zkwg_multi_create", "zkwg_multi_destroy", "zkwg_multi_devices",
    "zkwg_multi_circuit", "zkwg_calculate_batch_multi", "zkwg_calculate_batch_resident", "zkwg_resident_placement", "zkwg_resident_release",
    "zkwg_ntt_create", "zkwg_ntt_destroy", "zkwg_ntt_domain", "zkwg_ntt_work_bytes", "zkwg_ntt_transform_device", "zkwg_h_evaluations_device",
    "zkwg_msm_create", "zkwg_msm_destroy", "zkwg_msm_work_bytes", "zkwg_msm_window_bits", "zkwg_msm_g1_device",
    "zkwg_msm_g2_device", "zkwg_msm_create_g2", "zkwg_msm_create_device", "zkwg_msm_group", "zkwg_fixed_base_device", "zkwg_groth16_assemble", "zkwg_msm_enqueue_device", "zkwg_msm_finish_host", "zkwg_msm_create_ex", "zkwg_msm_work_bytes_batch", "zkwg_msm_lists_bytes", "zkwg_msm_estimate_work_bytes", "zkwg_msm_table_bytes", "zkwg_msm_precomputed", "zkwg_msm_enqueue_batch_device", "zkwg_msm_classify_device", "zkwg_msm_enqueue_lists_device", "zkwg_prover_emails_per_series", "zkwg_prover_contexts", "zkwg_prover_create_zkey", "zkwg_prover_create", "zkwg_prover_destroy", "zkwg_prover_prove_prepared", "zkwg_prover_prove_batch", "zkwg_device_alloc_chunked", "zkwg_device_free_chunked", "zkwg_device_alloc_chunked_ex",
]
