"""ctypes binding of lib/libschemanet_hip.so (C ABI: include/schemanet_hip.h).

This is the only place the product touches native code.  There is NO CPU fallback: if the
library is missing or no MI355X is visible, every entry point raises RuntimeError.

The reference binds its native code with pybind11 (`from .extension import feat_to_v_attr, ...`,
reference cpp_extension/__init__.py:5-10); here the binding is ctypes over plain pointers, with
torch used only to own device memory and provide the current HIP stream.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_void_p

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("SN_LIB_PATH") or os.path.join(_PKG, "lib", "libschemanet_hip.so")     # (override: kernel experiments of tools/)
ABI_VERSION = 12
# sha256[:16] of include/schemanet_hip.h with comments removed and whitespace collapsed: the declarations this binding (and
# ABI_VERSION) were written against.  tests/test_host_cpu.py::test_abi_version_names_the_header recomputes it, so a change to a
# signature or a struct without a new hash here - and, by the rule in the header, a new ABI_VERSION - fails the CPU suite.
ABI_HEADER_SHA = "cfb9f51fec2f16d9"
SN_MAX_TOKENS = 196
_lib = None


class _SizedArgs(Structure):
    """A by-pointer argument struct of the C ABI: its first member carries sizeof(the struct) as this binding lays it out;
    the library refuses a struct whose size is not its own (include/schemanet_hip.h, sn_abi_version)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.struct_size = ctypes.sizeof(type(self))


class RerankArgs(_SizedArgs):
    """struct sn_rerank_args (include/schemanet_hip.h)."""
    _fields_ = [
        ("struct_size", c_uint32),
        ("x", c_void_p), ("x_stride_b", c_int64), ("x_stride_l", c_int64), ("x_bf16", c_int),
        ("tok_stride_b", c_int64), ("tok_stride_l", c_int64), ("n_tokens", c_int64),
        ("codebook", c_void_p), ("packed", c_void_p), ("M", c_int), ("D", c_int),
        ("workspace", c_void_p), ("ids", c_void_p), ("ids_stride_b", c_int64), ("ids_stride_l", c_int64),
    ]


class GraphArgs(_SizedArgs):
    """struct sn_graph_args (include/schemanet_hip.h)."""
    _fields_ = [
        ("struct_size", c_uint32),
        ("ingredients", c_void_p), ("ing_stride_b", c_int64), ("ing_stride_l", c_int64),
        ("attn_cls", c_void_p), ("attn", c_void_p),
        ("acls_stride_b", c_int64), ("acls_stride_h", c_int64),
        ("attn_stride_b", c_int64), ("attn_stride_r", c_int64), ("attn_stride_h", c_int64),
        ("acls_heads", c_int), ("attn_heads", c_int),
        ("B", c_int), ("L", c_int),
        ("attn_cls_is_logits", c_int), ("attn_is_logits", c_int),
        ("use_clamp_v", c_int), ("use_clamp_e", c_int),
        ("clamp_v", c_float), ("clamp_e", c_float),
        ("geo", c_void_p), ("feat_h", c_int), ("feat_w", c_int),
        ("dist_alpha", c_float), ("dist_pow", c_float),
        ("w_v", c_void_p), ("w_e", c_void_p),
        ("mean", c_int), ("remove_self_loop", c_int),
        ("dict_keys", c_void_p), ("dict_vals", c_void_p), ("dict_off", c_void_p), ("dict_len", c_void_p),
        ("n_pad", c_int), ("pad_id", c_int64),
        ("out_ids", c_void_p), ("out_v2", c_void_p), ("out_v", c_void_p),
        ("out_e2", c_void_p), ("out_e", c_void_p),
        ("out_n", c_void_p), ("out_n_max", c_void_p), ("attn_cls_masked", c_void_p),
        ("skip_edge_padding", c_int),
        ("rerank", POINTER(RerankArgs)),
    ]


class GemmArgs(_SizedArgs):
    """struct sn_gemm_args (include/schemanet_hip.h)."""
    _fields_ = [
        ("struct_size", c_uint32),
        ("a_hi", c_void_p), ("a_lo", c_void_p), ("a_batch_stride", c_int64),
        ("b_hi", c_void_p), ("b_lo", c_void_p), ("b_batch_stride", c_int64),
        ("m", c_int), ("n", c_int), ("k", c_int), ("batches", c_int),
        ("c", c_void_p), ("c_batch_stride", c_int64), ("ldc", c_int),
        ("c_hi", c_void_p), ("c_lo", c_void_p), ("cp_batch_stride", c_int64), ("cp_cols", c_int),
        ("bias", c_void_p),
        ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("layernorm", c_int), ("relu", c_int),
        ("rows_valid", c_void_p),
        ("pool_w", c_void_p), ("pool_w_stride", c_int64), ("pooled", c_void_p),
        ("m_extent", c_void_p), ("k_extent", c_void_p),
        ("b_table_hi", c_void_p), ("b_table_lo", c_void_p),
        ("b_ids", c_void_p), ("b_ids_stride", c_int64), ("b_ids_n", c_int), ("b_table_rows", c_int),
        ("next_w_hi", c_void_p), ("next_w_lo", c_void_p),
        ("a_scale", c_void_p), ("b_scale", c_void_p), ("out_scale", c_void_p), ("next_w_scale", c_void_p), ("next_h_scale", c_void_p),
        ("extent_stride", c_int), ("accumulate", c_int), ("pooled_parts", c_int), ("zero_skipped", c_int),
    ]


_SIGNATURES = {
    "sn_abi_version": (c_int, []),
    "sn_last_error": (c_char_p, []),
    "sn_device_ok": (c_int, []),
    "sn_profile_enable": (c_int, [c_int]),
    "sn_profile_count": (c_int, [c_int]),
    "sn_profile_elapsed_ms": (c_int, [c_int, POINTER(c_float), c_int]),
    "sn_codebook_pack_bytes": (c_size_t, [c_int, c_int]),
    "sn_codebook_prepare": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sn_assign_variant": (c_int, []),
    "sn_assign_set_variant": (c_int, [c_int]),
    "sn_assign_workspace_bytes": (c_size_t, [c_int64]),
    "sn_assign_defers": (c_int, [c_int, c_int]),
    "sn_assign_words": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_int, c_void_p]),
    "sn_assign_words_bf16": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_int, c_void_p]),
    "sn_row_entropy": (c_int, [c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "sn_row_entropy_backward": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p]),
    "sn_kmeans_update": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "sn_kmeans_update_sorted": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int, c_int,
                                        c_void_p, c_void_p, c_void_p]),
    "sn_kmeans_distances": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                    c_int, c_int, c_void_p, c_void_p]),
    "sn_head_mean_attention": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_instance_graph": (c_int, [POINTER(GraphArgs), c_void_p]),
    "sn_full_vertices": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_limited_edges": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                 c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_int, c_int, c_void_p, c_int,
                                 c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_stats_accumulate": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_atlas_normalize": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_atlas_normalize_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "sn_atlas_normalize_entropy": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "sn_atlas_normalize_entropy_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "sn_gcn_adjacency": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sn_mask_layernorm_act": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "sn_weighted_pool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_layernorm_weighted_pool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                           c_void_p, c_void_p, c_void_p]),
    "sn_layernorm_split_planes": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p]),
    "sn_match_scores": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sn_class_votes": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sn_match_scores_votes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_pool_fc": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sn_pool_fc_t": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "sn_atlas_prune_rowsum": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_atlas_adjacency_planes": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_atlas_skip_pruned_rows": (None, [c_int]),
    "sn_atlas_keep_perm": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_class_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "sn_gcn_atlas_adjacency_planes_compact": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_plane_elems": (c_int64, [c_int, c_int]),
    "sn_gcn_adjacency_planes": (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_adjacency_planes_masked": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_adjacency_planes_per_graph": (c_int, [c_void_p, c_int, c_int, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_adjacency_planes_compact": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_sym_scatter_corner": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "sn_gcn_gather_planes": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_split_planes": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_split_planes_transposed": (c_int, [c_void_p, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_split_planes_nodes": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "sn_gcn_gemm": (c_int, [POINTER(GemmArgs), c_void_p]),
    "sn_pow2_scale_blocks": (c_int, [c_int64]),
    "sn_pow2_scale": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_sym_half_inplace": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "sn_normalize_sum_rows": (c_int, [c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "sn_ln_act_blocks": (c_int, [c_int64]),
    "sn_mask_layernorm_act_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "sn_mask_layernorm_act_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p,
                                               c_void_p, c_void_p, c_void_p]),
    "sn_embedding_grad_sorted": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sn_embedding_grad_scan": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sn_weighted_pool_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sn_graph_replace_memsets": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "sn_rectify_linear": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sn_weigh_blocks": (c_int, [c_int64]),
    "sn_weigh_attributes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "sn_weigh_attributes_backward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    # diagnostics (include/schemanet_hip.h, last section)
    "sn_debug_set_assign_options": (None, [c_int, c_int]),
    "sn_debug_screen_occupancy": (c_int, [c_int]),
    "sn_debug_set_stamps": (None, [c_void_p]),
    "sn_debug_set_graph_stamps": (None, [c_void_p]),
    "sn_debug_set_gemm_stamps": (None, [c_void_p]),
    "sn_debug_set_gemm_tile": (None, [c_int, c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def build(force=False):
    """hipcc-compile the library in-tree (cross-compiles for gfx950 without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _PKG, "lib/libschemanet_hip.so"])
    return LIB_PATH


def load():
    """dlopen the library and check the ABI; raises RuntimeError loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"schemanet HIP extension not built: {LIB_PATH} is missing (run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or `make -C schemanet-pytorch_amd`); there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libschemanet_hip.so ABI {lib.sn_abi_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def require_gpu():
    """The product path runs on an MI355X only."""
    if not torch.cuda.is_available():
        raise RuntimeError("schemanet HIP path needs a GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    lib = load()
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sn_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libschemanet_hip {what} failed ({rc}): {msg}")


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def to_device(t, device, dtype=None):
    """Plumbing: move an input to the compute device (the reference's callers hand CPU tensors
    to the extension, schema_net.py:314-315, 367-369)."""
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.device != device:
        t = t.to(device, non_blocking=True)
    return t


def compute_device(*tensors):
    """Device the kernels run on for a call: that of the first CUDA tensor, else cuda:current."""
    require_gpu()
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())
