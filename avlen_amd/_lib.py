"""ctypes binding of libavlen_hip.so (C ABI declared in include/avlen_hip.h).

There is NO fallback: if the HIP library is missing this module raises at import, and every call
checks the library's status code.  PyTorch only supplies device memory (``tensor.data_ptr()``) and the
current HIP stream.
"""
import ctypes as C
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libavlen_hip.so")

PREC_FP32, PREC_BF16, PREC_BF16X3, PREC_FP16 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_QUICKGELU = 0, 1, 2

_ERR = {1: "bad argument", 2: "kernel launch failed", 3: "workspace too small / missing"}


class AvlenHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950) or "
        "`python -c 'import __graft_entry__ as g; g.build()'`. avlen_amd has no CPU/PyTorch fallback.")

lib = C.CDLL(LIB_PATH)

f32p = C.c_void_p
vp = C.c_void_p


class Linear(C.Structure):
    _fields_ = [("w", f32p), ("b", f32p), ("out_f", C.c_int), ("in_f", C.c_int), ("w16", vp), ("ld16", C.c_int), ("w16lo", vp)]


class Conv(C.Structure):
    _fields_ = [("w", f32p), ("b", f32p), ("cin", C.c_int), ("cout", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int), ("w16", vp), ("cin16", C.c_int), ("w16c", vp), ("w16f", vp), ("w16lo", vp),
                ("w16flo", vp)]


class Affine(C.Structure):
    _fields_ = [("g", f32p), ("b", f32p)]


class ResBlock(C.Structure):
    _fields_ = [("conv1", Conv), ("conv2", Conv), ("down", Conv), ("bn1", Affine), ("bn2", Affine), ("bnd", Affine),
                ("has_down", C.c_int)]


class ResNet18(C.Structure):
    _fields_ = [("conv1", Conv), ("bn1", Affine), ("block", ResBlock * 8), ("fc", Linear)]


class Cnn3(C.Structure):
    _fields_ = [("conv", Conv * 3), ("fc", Linear), ("half_fmt", C.c_int)]


class Mha(C.Structure):
    _fields_ = [("in_proj", Linear), ("out_proj", Linear)]


class EncLayer(C.Structure):
    _fields_ = [("self_attn", Mha), ("lin1", Linear), ("lin2", Linear), ("norm1", Affine), ("norm2", Affine)]


class DecLayer(C.Structure):
    _fields_ = [("self_attn", Mha), ("cross_attn", Mha), ("lin1", Linear), ("lin2", Linear), ("norm1", Affine),
                ("norm2", Affine), ("norm3", Affine)]


class Transformer(C.Structure):
    _fields_ = [("enc", EncLayer), ("enc_norm", Affine), ("dec", DecLayer), ("dec_norm", Affine), ("d", C.c_int),
                ("nhead", C.c_int)]


class Smt(C.Structure):
    _fields_ = [("pose", Linear), ("fus0", Linear), ("fus2", Linear), ("tr", Transformer)]


class Dialog(C.Structure):
    _fields_ = [("fus0", Linear), ("fus2", Linear), ("tr", Transformer), ("pe", f32p), ("pe_len", C.c_int)]


class LnFold(C.Structure):
    _fields_ = [("w16f", vp), ("s", f32p), ("c", f32p)]


class ClipBlock(C.Structure):
    _fields_ = [("ln1", Affine), ("ln2", Affine), ("attn", Mha), ("fc", Linear), ("proj", Linear), ("attn_fold", LnFold),
                ("fc_fold", LnFold)]


class ClipText(C.Structure):
    _fields_ = [("tok_emb", f32p), ("pos_emb", f32p), ("block", ClipBlock * 12), ("ln_final", Affine),
                ("text_proj", f32p), ("vocab", C.c_int), ("ctx", C.c_int), ("width", C.c_int), ("heads", C.c_int),
                ("layers", C.c_int), ("out_dim", C.c_int), ("half_fmt", C.c_int), ("wstream", vp), ("text_proj_t", f32p)]


class Gru(C.Structure):
    _fields_ = [("w_ih", f32p), ("w_hh", f32p), ("b_ih", f32p), ("b_hh", f32p), ("in_f", C.c_int), ("hidden", C.c_int)]


class Heads(C.Structure):
    _fields_ = [("action", Linear), ("critic", Linear), ("unct", Linear), ("has_unct", C.c_int)]


class ExtMemOp(C.Structure):
    _fields_ = [("memory", C.c_void_p), ("masks", C.c_void_p), ("feats", C.c_void_p), ("ld_feats", C.c_int),
                ("not_done", C.c_void_p), ("masks_out", C.c_void_p), ("idx", C.c_int), ("total", C.c_int),
                ("capacity", C.c_int), ("N", C.c_int), ("dim", C.c_int)]


class Cmd(C.Structure):
    """avlen_cmd (csrc/sequencer.hip): one stream operation of a recorded launch sequence."""
    _fields_ = [("op", C.c_int), ("n", C.c_int), ("a", C.c_void_p), ("b", C.c_void_p), ("c", C.c_void_p), ("d", C.c_void_p)]


CMD_GRAPH, CMD_RECORD, CMD_WAIT, CMD_MULTICOPY = 1, 2, 3, 4

i32, f32, sz = C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/avlen_hip.h
SIGNATURES = {
    "avlen_gemm": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, sz, vp]),
    "avlen_gemm_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "avlen_gemm_pick_splitk": (i32, [i32, i32, i32]),
    "avlen_conv2d_nhwc": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "avlen_gemm_bf16": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_gemm_bf16_workspace_bytes": (sz, [i32, i32]),
    "avlen_conv2d_nhwc_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_conv_direct_bf16": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "avlen_cast_bf16": (i32, [vp, i32, vp, i32, C.c_long, i32, vp]),
    "avlen_cast_h16": (i32, [vp, i32, vp, i32, C.c_long, i32, i32, vp]),
    "avlen_gemm_tn_bf16_workspace_bytes": (sz, [C.c_long, i32, i32]),
    "avlen_cross1_expand": (i32, [vp, i32, vp, i32, vp, i32, vp]),
    "avlen_cross1_reduce": (i32, [vp, vp, i32, vp, vp, i32, i32, vp]),
    "avlen_cross1_dw": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "avlen_cross1_fwd": (i32, [vp, vp, C.c_long, vp, vp, vp, i32, i32, C.c_float, vp]),
    "avlen_cross1_bwd": (i32, [vp, vp, vp, vp, C.c_long, vp, vp, i32, i32, C.c_float, vp]),
    "avlen_gemm_tn_bf16": (i32, [vp, C.c_long, vp, C.c_long, C.c_long, i32, i32, vp, i32, C.c_float, vp, sz, vp]),
    "avlen_pack_conv_weight_h16": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "avlen_pack_fc_after_flatten_h16": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "avlen_resnet18_group_x3_workspace_bytes": (sz, [i32, i32]),
    "avlen_resnet18_group_fwd_x3": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "avlen_resnet18_group_fwd_x3_phase": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    "avlen_gemm_h16": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_pack_conv_weight_bf16": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "avlen_pack_conv_weight_frag": (i32, [vp, vp, i32, i32, vp]),
    "avlen_pack_fc_after_flatten_bf16": (i32, [vp, vp, i32, i32, i32, vp]),
    "avlen_pack_conv_weight": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "avlen_pack_fc_after_flatten": (i32, [vp, vp, i32, i32, i32, vp]),
    "avlen_groupnorm_nhwc": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp]),
    "avlen_layernorm_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]),
    "avlen_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "avlen_attention_fwd": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]),
    "avlen_attention_bwd": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, i32,
                                  i32, i32, i32, i32, i32, i32, f32, vp]),
    "avlen_attention_bwd_bf16": (i32, [vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, i32, vp, i32, vp, i32,
                                  i32, i32, i32, i32, i32, i32, f32, vp]),
    "avlen_preprocess_image": (i32, [vp, i32, vp, i32, i32, i32, f32, vp]),
    "avlen_rgbd_concat": (i32, [vp, i32, vp, vp, i32, i32, vp]),
    "avlen_feature_assemble": (i32, [vp, i32, C.POINTER(Linear), vp, i32, vp, i32, vp, i32, vp, i32, i32, vp, vp, vp,
                                     i32, i32, vp, i32, i32, vp, i32, i32, i32, vp]),
    "avlen_concat_rows": (i32, [vp, i32, i32, vp, i32, i32, vp, i32, i32, vp]),
    "avlen_resnet18_workspace_bytes": (sz, [i32]),
    "avlen_resnet18_fwd": (i32, [C.POINTER(ResNet18), vp, i32, i32, i32, i32, f32, vp, i32, i32, vp, sz, vp]),
    "avlen_resnet18_group_workspace_bytes": (sz, [i32, i32]),
    "avlen_resnet18_group_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_resnet18_any_workspace_bytes": (sz, [i32, i32, i32]),
    "avlen_resnet18_any_fwd": (i32, [C.POINTER(ResNet18), vp, i32, i32, i32, i32, vp, i32, i32, vp, sz, vp]),
    "avlen_resnet18_tv_workspace_bytes": (sz, [i32, i32, i32]),
    "avlen_resnet18_tv_fwd": (i32, [C.POINTER(ResNet18), vp, i32, i32, i32, i32, vp, i32, i32, vp, sz, vp]),
    "avlen_belief_input": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "avlen_belief_update": (i32, [vp, i32, vp, i32, vp, i32, vp, C.c_long, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32,
                                  i32, vp]),
    "avlen_resnet18_group_fwd_indexed": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "avlen_cnn3_fwd_indexed": (i32, [C.POINTER(Cnn3), vp, vp, i32, i32, i32, vp, i32, vp, sz, vp]),
    "avlen_cnn3_workspace_bytes": (sz, [C.POINTER(Cnn3), i32, i32, i32]),
    "avlen_cnn3_fwd": (i32, [C.POINTER(Cnn3), vp, i32, i32, i32, vp, i32, i32, vp, sz, vp]),
    "avlen_cnn3_group_workspace_bytes": (sz, [C.POINTER(Cnn3), i32, i32, i32, i32]),
    "avlen_cnn3_group_fwd": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    "avlen_smt_workspace_bytes": (sz, [C.POINTER(Smt), i32, i32, i32, i32]),
    "avlen_smt_fwd": (i32, [C.POINTER(Smt), vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_smt_bwd": (i32, [C.POINTER(Smt), C.POINTER(Smt), vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    "avlen_dialog_train_workspace_bytes": (sz, [C.POINTER(Dialog), i32, i32]),
    "avlen_dialog_train_fwd": (i32, [C.POINTER(Dialog), vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "avlen_dialog_bwd": (i32, [C.POINTER(Dialog), C.POINTER(Dialog), vp, vp, i32, vp, vp, i32, i32, i32, vp, sz, vp]),
    "avlen_linear_bwd_workspace_bytes": (sz, []),
    "avlen_linear_bwd": (i32, [C.POINTER(Linear), C.POINTER(Linear), vp, i32, vp, i32, vp, i32, i32, i32, vp, sz, vp]),
    "avlen_action_encoder_bwd": (i32, [vp, i32, vp, C.POINTER(Linear), i32, vp]),
    "avlen_dialog_loss_heads_bwd": (i32, [C.POINTER(Heads), C.POINTER(Heads), vp, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "avlen_cnn3_train_workspace_bytes": (sz, [C.POINTER(Cnn3), i32, i32, i32, i32]),
    "avlen_cnn3_train_fwd": (i32, [C.POINTER(Cnn3), vp, i32, i32, i32, vp, i32, i32, vp, sz, vp]),
    "avlen_cnn3_train_bwd": (i32, [C.POINTER(Cnn3), C.POINTER(Cnn3), vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_dialog_workspace_bytes": (sz, [C.POINTER(Dialog), i32, i32]),
    "avlen_dialog_fwd": (i32, [C.POINTER(Dialog), vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "avlen_clip_text_workspace_bytes": (sz, [C.POINTER(ClipText), i32]),
    "avlen_clip_text_fwd": (i32, [C.POINTER(ClipText), vp, vp, i32, i32, vp, sz, vp]),
    "avlen_clip_text_cache_bytes": (sz, [C.POINTER(ClipText), i32]),
    "avlen_clip_text_cached_fwd": (i32, [C.POINTER(ClipText), vp, vp, sz, vp, i32, i32, vp, sz, vp]),
    "avlen_clip_text_dialog_fwd": (i32, [C.POINTER(ClipText), C.POINTER(Linear), vp, vp, sz, vp, i32, i32, vp, sz, vp, vp, i32, vp]),
    "avlen_clip_stream_bytes": (sz, [C.POINTER(ClipText)]),
    "avlen_clip_pack_stream": (i32, [C.POINTER(ClipText), vp, i32, vp]),
    "avlen_gru_workspace_bytes": (sz, [C.POINTER(Gru), i32, i32]),
    "avlen_gru_fwd": (i32, [C.POINTER(Gru), vp, vp, vp, vp, vp, i32, i32, i32, vp, sz, vp]),
    "avlen_heads_fwd": (i32, [C.POINTER(Heads), vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "avlen_sample_race": (i32, [vp, vp, vp, i32, i32, vp]),
    "avlen_heads_act_fwd": (i32, [C.POINTER(Heads), vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "avlen_heads_act_host_fwd": (i32, [C.POINTER(Heads), vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "avlen_ppo_loss_heads_bwd": (i32, [C.POINTER(Heads), C.POINTER(Heads), vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp,
                                       f32, f32, f32, f32, vp, vp, i32, vp]),
    "avlen_rl_mask_norm": (i32, [vp, i32, vp, vp]),
    "avlen_gae_scan": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp]),
    "avlen_resnet18_train_workspace_bytes": (sz, [C.POINTER(ResNet18), i32, i32, i32, i32]),
    "avlen_resnet18_train_fwd": (i32, [C.POINTER(ResNet18), vp, i32, i32, i32, vp, i32, i32, vp, sz, vp]),
    "avlen_resnet18_train_bwd": (i32, [C.POINTER(ResNet18), C.POINTER(ResNet18), vp, vp, i32, i32, i32, i32, vp, i32, vp, sz, vp]),
    "avlen_belief_regression_loss": (i32, [vp, vp, C.c_long, vp, i32, vp, vp, i32, vp]),
    "avlen_spectrogram_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "avlen_spectrogram": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "avlen_discounted_returns": (i32, [vp, vp, vp, vp, i32, i32, f32, vp]),
    "avlen_baseline_train_workspace_bytes": (sz, [C.POINTER(Cnn3), C.POINTER(Cnn3), C.POINTER(Gru), i32, i32, i32, i32, i32, i32]),
    "avlen_baseline_train_fwd": (i32, [C.POINTER(Cnn3), C.POINTER(Cnn3), C.POINTER(Gru), vp, vp, i32, vp, vp, i32, vp, vp, vp, vp,
                                       i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_baseline_train_bwd": (i32, [C.POINTER(Cnn3), C.POINTER(Cnn3), C.POINTER(Gru), C.POINTER(Cnn3), C.POINTER(Cnn3),
                                       C.POINTER(Gru), vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "avlen_grad_sumsq": (i32, [vp, sz, vp, vp]),
    "avlen_adam_step": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, f32, i32, f32, vp, vp]),
    "avlen_extmem_insert": (i32, [vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "avlen_extmem_insert_multi": (i32, [vp, i32, vp]),
    "avlen_set_big_m": (None, [C.c_long]),
    "avlen_set_x3_mixed_backward_rows": (None, [C.c_long]),
    "avlen_set_big16": (None, [i32]),
    "avlen_set_audio3": (None, [i32]),
    "avlen_set_tower_x3_reserved_cus": (None, [i32]),
    "avlen_tower_x3_timing": (i32, [vp, vp, i32]),
    "avlen_set_clip_tower_split4_wgs": (None, [i32]),
    "avlen_set_chain_one_xcd": (None, [i32]),
    "avlen_minibatch_gather": (i32, [vp, vp, vp, i32, i32, i32, sz, i32, vp]),
    "avlen_copy_rows": (i32, [vp, i32, vp, i32, i32, i32, vp]),
    "avlen_gather_rows": (i32, [vp, i32, vp, vp, i32, i32, i32, vp]),
    "avlen_multi_copy": (i32, [vp, vp, vp, i32, vp]),
    "avlen_comm_unique_id": (i32, [vp, sz]),
    "avlen_comm_init_rank": (i32, [vp, i32, vp, i32]),
    "avlen_comm_destroy": (i32, [vp]),
    "avlen_grad_allreduce": (i32, [vp, sz, i32, vp, vp]),
    "avlen_cmds_run": (i32, [vp, i32]),
    "avlen_prefetch_l2": (i32, [vp, vp, i32, vp]),
    "avlen_ln_fold_weights": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp]),
    "avlen_ln_fold_weights_h16": (i32, [vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp]),
    "avlen_build_info": (C.c_char_p, []),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)            # AttributeError here == missing export: fail loudly
    _fn.restype, _fn.argtypes = _res, _args


def stream():
    """The current torch HIP stream as a raw hipStream_t (torch.cuda.current_stream() builds a Stream object and costs
    several microseconds per call -- a dozen calls per rollout step; this path stays well under one)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "avlen_hip needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def check(rc, what=""):
    if rc != 0:
        raise AvlenHipError(f"avlen_hip call failed ({what}): {_ERR.get(rc, rc)}")


def call(name, *args):
    check(getattr(lib, name)(*args), name)


def multi_copy(pairs):
    """pairs: iterable of (dst, src) tensors.  Device-resident, contiguous, same-dtype, same-size pairs go out as ONE
    `avlen_multi_copy` launch on the current stream; anything else (host tensors, dtype conversion, strided views)
    falls back to `Tensor.copy_`."""
    batch = []
    for dst, src in pairs:
        if (torch.is_tensor(src) and src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and src.is_contiguous()
                and dst.is_contiguous() and src.numel() == dst.numel()):
            if src.data_ptr() != dst.data_ptr() and src.numel():
                batch.append((dst, src))
        else:
            src = src if torch.is_tensor(src) else torch.as_tensor(src)
            if dst.dtype == torch.uint8 and src.is_floating_point():
                # a float frame going into uint8 storage (RolloutStorage(uint8_sensors=...)): round to nearest and saturate -- a plain
                # cast truncates fractions and wraps values outside 0..255 (e.g. an augmented or 0..1-normalised image)
                src = src.round().clamp_(0, 255)
            dst.copy_(src, non_blocking=True)
    if not batch:
        return
    n = len(batch)
    srcs = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in batch])
    dsts = (C.c_void_p * n)(*[d.data_ptr() for d, _ in batch])
    sizes = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in batch])
    call("avlen_multi_copy", srcs, dsts, sizes, n, stream())
