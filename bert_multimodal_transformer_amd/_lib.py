"""ctypes binding of libmagbert_hip.so (include/magbert_hip.h).  The HIP library IS the product: importing the
package without it (or calling any operator without a GPU) raises -- there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MB_LIB_DIR: another build of the same library (scripts/build_variant.py -> gpurun_ab/<name>/) for same-box A/B measurements
LIB_PATH = os.path.join(os.environ.get("MB_LIB_DIR") or os.path.join(_HERE, "lib"), "libmagbert_hip.so")

DT_F32, DT_BF16 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RES, EPI_ADD_RES, EPI_DGELU, EPI_ACCUM_F32, EPI_BIAS_F32 = range(7)


class DropKey(C.Structure):
    _fields_ = [("k0", C.c_uint32), ("k1", C.c_uint32), ("thresh", C.c_uint32), ("scale", C.c_float)]


class BertEngineConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("intermediate_size", C.c_int), ("max_position", C.c_int), ("type_vocab", C.c_int),
                ("num_labels", C.c_int), ("visual_dim", C.c_int), ("acoustic_dim", C.c_int), ("pad_token_id", C.c_int),
                ("layer_norm_eps", C.c_float), ("mag_layer_norm_eps", C.c_float), ("beta_shift", C.c_float),
                ("hidden_dropout", C.c_float), ("attn_dropout", C.c_float), ("mag_dropout", C.c_float),
                ("dtype", C.c_int), ("max_batch", C.c_int), ("max_seq", C.c_int)]


class XlnetEngineConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("d_model", C.c_int), ("n_layer", C.c_int), ("n_head", C.c_int), ("d_inner", C.c_int),
                ("num_labels", C.c_int), ("visual_dim", C.c_int), ("acoustic_dim", C.c_int), ("injection_index", C.c_int),
                ("layer_norm_eps", C.c_float), ("mag_layer_norm_eps", C.c_float), ("beta_shift", C.c_float),
                ("dropout", C.c_float), ("summary_last_dropout", C.c_float), ("mag_dropout", C.c_float),
                ("dtype", C.c_int), ("max_batch", C.c_int), ("max_seq", C.c_int)]


_vp, _i, _f, _sz, _u64, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_uint64, C.c_uint32
_dk = C.POINTER(DropKey)

# name -> (restype, argtypes): every symbol include/magbert_hip.h declares
PROTOTYPES = {
    "mb_error_string": (C.c_char_p, [_i]),
    "mb_version": (_i, []),
    "mb_make_dropkey": (None, [_u64, _u64, _u32, _f, _dk]),
    "mb_gemm": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _dk, _i, _i, _vp]),
    "mb_gemm_grouped_wgrad": (_i, [_i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mb_narrow": (_i, [_i, _vp, _vp, _sz, _vp]),
    "mb_widen": (_i, [_i, _vp, _vp, _sz, _vp]),
    "mb_layernorm_forward": (_i, [_i, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _dk, _vp]),
    "mb_layernorm_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _dk, _dk, _vp]),
    "mb_embed_forward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _dk, _vp]),
    "mb_embed_backward": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                               _i, _dk, _vp]),
    "mb_attention_forward": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _dk, _vp]),
    "mb_attention_backward": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _dk, _vp]),
    "mb_mag_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mb_mag_forward": (_i, [_i] + [_vp] * 13 + [_f, _dk, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mb_mag_backward": (_i, [_i] + [_vp] * 11 + [_f, _dk] + [_vp] * 14 + [_i, _i, _i, _i, _vp]),
    "mb_adamw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _sz, _sz, _sz, _f, _f, _f, _f, _f, _i, _i, _f, _i, _vp]),
    "mb_bert_create": (_i, [C.POINTER(BertEngineConfig), C.POINTER(_vp)]),
    "mb_bert_destroy": (None, [_vp]),
    "mb_bert_num_tensors": (_i, [_vp]),
    "mb_bert_tensor_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i),
                                 C.POINTER(C.c_int64), C.POINTER(_i)]),
    "mb_bert_param_count": (_sz, [_vp]),
    "mb_bert_decay_count": (_sz, [_vp]),
    "mb_bert_shadow_range": (None, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mb_bert_workspace_bytes": (_sz, [_vp]),
    "mb_bert_bind": (_i, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "mb_bert_sync_weights": (_i, [_vp, _vp]),
    "mb_bert_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp]),
    "mb_bert_backward": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "mb_bert_sequence_output": (_vp, [_vp]),
    "mb_bert_pooled_output": (_vp, [_vp]),
    "mb_bert_hidden_state": (_vp, [_vp, _i]),
    "mb_bert_set_attention_output": (_i, [_vp, _vp]),
    "mb_debug_gemm_trace": (_i, [_vp, _i]),
    "mb_debug_attention_trace": (_i, [_vp, _i]),
    "mb_xlnet_attention_probs": (_vp, [_vp, _i, C.POINTER(_i)]),
    "mb_xlnet_set_head_mask": (_i, [_vp, _vp]),
    "mb_xlnet_set_perm_mask": (_i, [_vp, _vp]),
    "mb_xlnet_set_mems": (_i, [_vp, _vp, _i]),
    "mb_xlnet_query_stream_scratch_bytes": (_sz, [_vp, _i, _i, _i]),
    "mb_xlnet_query_stream_state_bytes": (_sz, [_vp, _i, _i]),
    "mb_xlnet_query_stream": (_i, [_vp, _vp, _i, _vp, _sz, _vp, _vp]),
    "mb_bert_mark_grads_zero": (_i, [_vp, _i]),
    "mb_xlnet_mark_grads_zero": (_i, [_vp, _i]),
    "mb_bert_materialize_grads": (_i, [_vp, _vp]),
    "mb_xlnet_materialize_grads": (_i, [_vp, _vp]),
    "mb_bert_grads_stale": (_i, [_vp]),
    "mb_xlnet_grads_stale": (_i, [_vp]),
    "mb_bert_set_head_mask": (_i, [_vp, _vp]),
    "mb_bert_set_inputs_embeds": (_i, [_vp, _vp]),
    "mb_bert_set_position_ids": (_i, [_vp, _vp]),
    "mb_bert_inputs_embeds_grad": (_vp, [_vp]),
    "mb_bert_backward_outputs": (_i, [_vp, _vp, _vp, _vp]),
    "mb_bert_stage_grad_ranges": (_i, [_vp, _i, C.POINTER(_sz), C.POINTER(_sz), _i]),
    "mb_bert_train_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f,
                                _i, _i, _f, _f, _i, _vp]),
    "mb_bert_stage_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _vp, _vp, _i, _vp]),
    "mb_bert_stage_backward": (_i, [_vp, _f, _i, _i, _vp]),
    "mb_bert_staged_input_ids": (_vp, [_vp]),
    "mb_bert_load_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_vp), _vp]),
    "mb_bert_graph_stats": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mb_bert_set_profiling": (_i, [_vp, _i]),
    "mb_bert_profile_wgrad_us": (_i, [_vp, C.POINTER(_f)]),
    "mb_bert_profile_adamw_us": (_i, [_vp, C.POINTER(_f)]),
    "mb_xlnet_set_inputs_embeds": (_i, [_vp, _vp]),
    "mb_xlnet_inputs_embeds_grad": (_vp, [_vp]),
    "mb_xlnet_model_output": (_vp, [_vp, _vp]),
    "mb_xlnet_backward_outputs": (_i, [_vp, _vp, _vp]),
    "mb_xlnet_set_profiling": (_i, [_vp, _i]),
    "mb_xlnet_profile_wgrad_us": (_i, [_vp, C.POINTER(_f)]),
    "mb_xlnet_profile_adamw_us": (_i, [_vp, C.POINTER(_f)]),
    "mb_xlnet_create": (_i, [C.POINTER(XlnetEngineConfig), C.POINTER(_vp)]),
    "mb_xlnet_destroy": (None, [_vp]),
    "mb_xlnet_num_tensors": (_i, [_vp]),
    "mb_xlnet_tensor_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_i),
                                  C.POINTER(C.c_int64), C.POINTER(_i)]),
    "mb_xlnet_param_count": (_sz, [_vp]),
    "mb_xlnet_decay_count": (_sz, [_vp]),
    "mb_xlnet_shadow_range": (None, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mb_xlnet_workspace_bytes": (_sz, [_vp]),
    "mb_xlnet_bind": (_i, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "mb_xlnet_sync_weights": (_i, [_vp, _vp]),
    "mb_xlnet_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp]),
    "mb_xlnet_backward": (_i, [_vp, _vp, _vp, _f, _i, _i, _vp]),
    "mb_xlnet_sequence_output": (_vp, [_vp]),
    "mb_xlnet_hidden_state": (_vp, [_vp, _i]),
    "mb_xlnet_train_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f,
                                 _i, _i, _f, _f, _i, _vp]),
    "mb_xlnet_load_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.POINTER(_vp), _vp]),
    "mb_xlnet_graph_stats": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mb_xlnet_trainable_count": (_sz, [_vp]),
    "mb_xlnet_stage_grad_ranges": (_i, [_vp, _i, C.POINTER(_sz), C.POINTER(_sz), _i]),
    # data parallel (csrc/comm.hip)
    "mb_comm_unique_id": (_i, [_vp]),
    "mb_comm_create_rccl": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "mb_comm_create_callbacks": (_i, [_i, _i, _vp, _vp, _vp, C.POINTER(_vp)]),
    "mb_comm_destroy": (None, [_vp]),
    "mb_comm_rank": (_i, [_vp]),
    "mb_comm_world": (_i, [_vp]),
    "mb_comm_stream": (_vp, [_vp]),
    "mb_comm_scratch_bytes": (_sz, [_i, _i, _sz, _i, _i, _i]),
    "mb_comm_bind_scratch": (_i, [_vp, _vp, _sz, _i, _sz, _i, _i, _i]),
    "mb_comm_all_reduce": (_i, [_vp, _vp, _sz, _vp]),
    "mb_comm_exchange_rows": (_i, [_vp, _vp, _vp, _i, _vp]),
    "mb_comm_set_row_exchange": (_i, [_vp, _i]),
    "mb_comm_set_sharding": (_i, [_vp, _i]),
    "mb_comm_sharding": (_i, [_vp]),
    "mb_comm_join": (_i, [_vp, _vp]),
    "mb_comm_gather_shards": (_i, [_vp, _vp, _i, _vp]),
    "mb_comm_shard_slices": (_i, [_vp, C.POINTER(_sz), _i]),
    "mb_comm_set_timing": (_i, [_vp, _i]),
    "mb_comm_exposed_ms": (_i, [_vp, C.POINTER(_f)]),
    "mb_comm_stats": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "mb_comm_last_error": (C.c_char_p, []),
    "mb_bert_train_step_dp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f,
                                   _i, _i, _f, _f, _i, _vp, _vp]),
    "mb_xlnet_train_step_dp": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f,
                                    _i, _i, _f, _f, _i, _vp, _vp]),
}

ALL_REDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
ALL_GATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_lib = None


class MagbertError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises (never falls back) when the .so is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MagbertError(
                "libmagbert_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(h, name)        # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code):
    if code != 0:
        msg = lib().mb_error_string(code).decode()
        if code == 1005 or 2000 <= code < 2100:
            msg += ": " + (lib().mb_comm_last_error() or b"").decode()
        raise MagbertError("%s (code %d)" % (msg, code))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def make_dropkey(seed, step, site, p):
    k = DropKey()
    lib().mb_make_dropkey(int(seed) & (2**64 - 1), int(step), int(site), float(p), C.byref(k))
    return k


def no_drop():
    return DropKey(0, 0, 0, 1.0)
