"""ctypes binding of libsketchycolor_hip.so (C ABI in include/sketchycolor_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call
fails, this module raises.  PyTorch is used only to own device memory and the
stream; kernels receive raw device pointers.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SSC_LIB_PATH') or os.path.join(_HERE, 'lib', 'libsketchycolor_hip.so')

ACT_PRELU = 5       # concat_parts only: the part's ab is the scalar leak
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_MIU = 0, 1, 2, 3, 4     # TANH / MIU: pointwise kernels only, never on load


class GView(C.Structure):
    _fields_ = [('s0', C.c_void_p), ('s1', C.c_void_p), ('ab0', C.c_void_p), ('ab1', C.c_void_p),
                ('C0', C.c_int32), ('C1', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('act', C.c_int32), ('act1', C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [('x', GView), ('w', C.c_void_p), ('bias', C.c_void_p), ('out', C.c_void_p),
                ('NB', C.c_int32), ('PH', C.c_int32), ('PW', C.c_int32),
                ('TH', C.c_int32), ('TW', C.c_int32), ('in_stride', C.c_int32),
                ('ioff_y', C.c_int32), ('ioff_x', C.c_int32), ('nphase', C.c_int32),
                ('ky0', C.c_int32), ('kx0', C.c_int32), ('kstep', C.c_int32),
                ('KH', C.c_int32), ('KW', C.c_int32), ('wC0', C.c_int32), ('wC1', C.c_int32),
                ('bmode', C.c_int32), ('k_real', C.c_int32), ('n_off', C.c_int32), ('Nn', C.c_int32),
                ('Nstore', C.c_int32), ('OH', C.c_int32), ('OW', C.c_int32), ('ldc', C.c_int32),
                ('out_stride', C.c_int32), ('ooff_y', C.c_int32), ('ooff_x', C.c_int32),
                ('epi', C.c_int32), ('accumulate', C.c_int32), ('stat_partial', C.c_void_p), ('sk_flags', C.c_void_p),
                ('sb_x', C.c_void_p), ('sb_ab', C.c_void_p), ('sb_stats', C.c_void_p), ('sb_ldx', C.c_int32),
                ('sb_act', C.c_int32), ('sk_tag', C.c_int32), ('sb2_col0', C.c_int32),
                ('sb2_x', C.c_void_p), ('sb2_ab', C.c_void_p), ('sb2_stats', C.c_void_p), ('stat_partial2', C.c_void_p),
                ('sb2_ldx', C.c_int32), ('sb2_act', C.c_int32),
                ('stat_mode', C.c_int32), ('ws_kc', C.c_int32),
                ('wsplit', C.c_void_p), ('ws_nbp', C.c_int32), ('lds_hint', C.c_int32)]


class SplitJob(C.Structure):
    _fields_ = [('w', C.c_void_p), ('dst', C.c_void_p), ('taps', C.c_int32), ('c0', C.c_int32), ('c1', C.c_int32),
                ('orient', C.c_int32), ('first_thread', C.c_int64)]


class BnApplyJob(C.Structure):
    _fields_ = [('x', C.c_void_p), ('M', C.c_int64), ('C', C.c_int32), ('ldx', C.c_int32), ('ab', C.c_void_p),
                ('stats', C.c_void_p), ('g1', C.c_void_p), ('g2', C.c_void_p), ('ldg1', C.c_int32), ('act1', C.c_int32),
                ('ldg2', C.c_int32), ('act2', C.c_int32), ('has_bn', C.c_int32), ('rowb_P', C.c_int32), ('rowb', C.c_void_p),
                ('rowb_scale', C.c_float), ('lddx', C.c_int32), ('coef', C.c_void_p), ('dx', C.c_void_p)]


class BnBwdSite(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ab', C.c_void_p), ('stats', C.c_void_p), ('partial', C.c_void_p),
                ('partial_bytes', C.c_int64), ('ldx', C.c_int32), ('act', C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [('g', GView), ('d', GView), ('out', C.c_void_p),
                ('NB', C.c_int32), ('PH', C.c_int32), ('PW', C.c_int32),
                ('TH', C.c_int32), ('TW', C.c_int32), ('in_stride', C.c_int32),
                ('ioff_y', C.c_int32), ('ioff_x', C.c_int32),
                ('Cg_real', C.c_int32), ('Nn', C.c_int32), ('ldc', C.c_int32), ('accumulate', C.c_int32),
                ('exact', C.c_int32), ('_pad', C.c_int32)]


_lib = None


def _dev_env(name, default=None):
    """Developer A/B switches are honoured only under SSC_DEV_SWITCHES=1 (lab tools, not supported configurations; DESIGN.md
    section 7 lists the supported switches)."""
    return os.environ.get(name, default) if os.environ.get('SSC_DEV_SWITCHES') == '1' else default


def lib():
    """Load the HIP library; raise loudly when it is not there (no CPU fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libsketchycolor_hip.so is missing (%s): run `python -c "import __graft_entry__ as g; '
                               'g.build()"` or `python -m sketchyscenecolorization_amd.build`' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        _declare(l)
        check_build_hash(l)
        _lib = l
    return _lib


def build_hash(l=None):
    """The source hash compiled into the loaded library (ssc_build_hash)."""
    buf = C.create_string_buffer(32)
    (l if l is not None else lib()).ssc_build_hash(buf, 32)
    return buf.value.decode()


def check_build_hash(l=None, tree=None):
    """Raise when the library was built from other kernel sources than the tree it is used from (a stale binary would run --
    and be profiled, benchmarked, tested -- under the name of sources it does not contain).  SSC_ALLOW_STALE_LIB=1 turns the
    error into a warning (A/B runs of an older build against new host code)."""
    from . import build
    have = build_hash(l)
    if tree is None:
        try:
            tree = build.tree_hash()
        except OSError:         # a deployment that ships the library without csrc/ or include/: nothing to compare with
            import warnings
            warnings.warn('%s: kernel sources not found beside the package, build hash %s not checked' % (LIB_PATH, have))
            return have
    want = tree
    if have != want:
        msg = ('%s was built from kernel sources %s, the tree is %s: rebuild (python -m sketchyscenecolorization_amd.build)'
               % (LIB_PATH, have, want))
        if os.environ.get('SSC_ALLOW_STALE_LIB') == '1':
            import warnings
            warnings.warn(msg)
        else:
            raise RuntimeError(msg)
    return have


_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float

# name -> argtypes (restype is always int); mirrors include/sketchycolor_hip.h
SIGNATURES = {
    'ssc_version': [],
    'ssc_build_hash': [C.c_char_p, _I],
    'ssc_device_info': [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, _I],
    'ssc_conv_forward': [C.POINTER(ConvDesc), _P, _L, _P],
    'ssc_filter_split_geom': [_I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    'ssc_filter_split': [_P, _I, _I, _I, _I, _P, _P],
    'ssc_filter_split_batch': [_P, _I, _L, _P],
    'ssc_bf16_prepare': [],
    'ssc_conv_wgrad': [C.POINTER(WgradDesc), _P, _L, _P],
    'ssc_conv_wgrad128_supported': [C.POINTER(WgradDesc)],
    'ssc_conv_wgn16_supported': [C.POINTER(WgradDesc)],
    'ssc_conv_wgrad128': [C.POINTER(WgradDesc), _P, _L, _P],
    'ssc_conv_forward_bn': [C.POINTER(ConvDesc), _P, _L, _P, _P, _F, _P, _P, _P],
    'ssc_bn_finalize': [_P, _I, _I, _L, _P, _P, _F, _P, _P, _P],
    'ssc_conv_forward_bnbwd': [C.POINTER(ConvDesc), _P, _L, _P, _I, _P, _P, _I, _P, _L, C.POINTER(C.c_int), _P],
    'ssc_conv_forward_minmax': [C.POINTER(ConvDesc), _P, _L, _P, _P],
    'ssc_minmax_finalize': [_P, _I, _I, _I, _P, _P],
    'ssc_conv_forward_bnbwd2': [C.POINTER(ConvDesc), _P, _L, C.POINTER(BnBwdSite), C.POINTER(BnBwdSite), _I,
                                C.POINTER(C.c_int), _P],
    'ssc_bn_act_backward_pre': [_P, _L, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _F, _I, _P,
                                _L, _P],
    'ssc_head1_forward_supported': [C.POINTER(ConvDesc)],
    'ssc_head1_forward': [C.POINTER(ConvDesc), _P, _L, _P],
    'ssc_head1_dgrad_supported': [C.POINTER(ConvDesc)],
    'ssc_head1_dgrad': [C.POINTER(ConvDesc), _P],
    'ssc_head1_wgrad_supported': [C.POINTER(WgradDesc)],
    'ssc_head1_wgrad': [C.POINTER(WgradDesc), _P, _L, _P],
    'ssc_head1_dgrad_bn_backward': [C.POINTER(ConvDesc), _P, _P, _P, _I, _P, _F, _P, _P, _P, _P, _L, _P],
    'ssc_bn_bwd_finalize': [_P, _I, _I, _L, _P, _P, _P, _P],
    'ssc_block_out_backward': [_P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P],
    'ssc_bn_bwd_sums': [C.POINTER(BnApplyJob), _P, _I, _P, _P, _P, _P, _L, _P],
    'ssc_bn_bwd_apply': [C.POINTER(BnApplyJob), _P],
    'ssc_conv_narrow_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_narrow_forward': [C.POINTER(ConvDesc), _P],
    'ssc_conv_fewchan_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_pw1x1_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_c3x3_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_s2n16_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_tr4n16_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_fewchan7_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_tr4_tiny_supported': [C.POINTER(ConvDesc)],
    'ssc_conv_forward_kernel_name': [C.POINTER(ConvDesc), C.c_char_p, _I],
    'ssc_conv_wgrad_kernel_name': [C.POINTER(WgradDesc), C.c_char_p, _I],
    'ssc_conv_forward_plan': [C.POINTER(ConvDesc), _L, C.POINTER(C.c_int)],
    'ssc_sk_configure': [_I, _I],
    'ssc_nchw_to_nhwc': [_P, _P, _I, _I, _I, _I, _I, _P],
    'ssc_nhwc_to_nchw': [_P, _P, _I, _I, _I, _I, _I, _P],
    'ssc_sketch_preprocess_u8': [_P, _I, _I, _I, _I, _P, _P],
    'ssc_image_postprocess_u8': [_P, _I, _I, _L, _P, _P],
    'ssc_resample_u8': [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'ssc_decode_paired_u8': [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    'ssc_distance_map_u8': [_P, _I, _I, _P, _P, _L, _P],
    'ssc_fill': [_P, _F, _L, _P],
    'ssc_timestamp': [_P, _P],
    'ssc_affine_act': [_P, _I, _P, _I, _I, _P, _I, _L, _I, _P],
    'ssc_residual_merge': [_P, _P, _P, _P, _I, _P, _L, _I, _P],
    'ssc_bn_stats': [_P, _L, _I, _I, _P, _P, _F, _P, _P, _P, _L, _P],
    'ssc_bn_act_backward': [_P, _L, _I, _I, _P, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _L, _P],
    'ssc_embedding_gather': [_P, _P, _I, _I, _P, _P],
    'ssc_embedding_scatter_add': [_P, _I, _P, _I, _I, _P, _P],
    'ssc_row_l2norm_fwd': [_P, _I, _P, _L, _I, _P, _P, _P],
    'ssc_row_l2norm_bwd': [_P, _P, _P, _L, _I, _P, _I, _P],
    'ssc_lstm_pointwise_fwd': [_P, _P, _P, _I, _P, _I, _P, _P, _L, _I, _P, _P, _P, _P],
    'ssc_lstm_step_fwd': [_P, _P, _I, _P, _P, _I, _P, _I, _P, _L, _I, _I, _P, _P, _P, _P],
    'ssc_lstm_step_fwd_bf': [_P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _L, _I, _P, _P, _P, _P, _P],
    'ssc_lstm_hsplit': [_P, _L, _I, _P, _P],
    'ssc_lstm_pointwise_bwd': [_P, _P, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _P],
    'ssc_squash_fwd': [_P, _L, _P, _P],
    'ssc_squash_bwd': [_P, _P, _P, _L, _P, _P],
    'ssc_group_rowsum': [_P, _I, _L, _I, _I, _P, _I, _P],
    'ssc_act_mean_hw': [_P, _P, _I, _I, _I, _I, _P, _P],
    'ssc_add_row_bcast': [_P, _P, _F, _I, _I, _I, _P],
    'ssc_miu_permute_fwd': [_P, _I, _I, _I, _P, _P],
    'ssc_miu_permute_bwd': [_P, _P, _I, _I, _I, _P, _P],
    'ssc_softplus_loss': [_P, _I, _L, _F, _F, _P, _P, _F, _P],
    'ssc_acgan_loss': [_P, _P, _I, _I, _I, _F, _P, _P, _P],
    'ssc_gen_output_grad': [_P, _I, _P, _I, _P, _I, _L, _F, _P, _P, _P],
    'ssc_l2_reg': [_P, _L, _F, _P, _P, _P],
    'ssc_adam_tf': [_P, _P, _P, _P, _L, _F, _P, _F, _F, _F, _F, _P],
    'ssc_sn_forward': [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    'ssc_sn_backward': [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P],
    'ssc_axpy': [_P, _P, _F, _L, _P],
    'ssc_optimizer_step': [_I, _P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P],
    'ssc_bg_gan_loss': [_P, _L, _I, _F, _P, _P, _F, _P],
    'ssc_count_nonzero_i32': [_P, _L, _P, _P, _L, _P],
    'ssc_bg_output_grad': [_P, _P, _P, _P, _F, _P, _P, _P, _L, _P],
    'ssc_seg_ce_loss': [_P, _I, _P, _L, _F, _P, _P, _I, _P],
    'ssc_sn_forward_any': [_P, _P, _I, _I, _P, _P, _P, _P, _P, _L, _P],
    'ssc_sn_backward_any': [_P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _L, _P],
    'ssc_mean_pool2': [_P, _I, _P, _I, _I, _I, _I, _I, _P],
    'ssc_cbn_fold': [_P, _P, _P, _P, _I, _I, _P, _P],
    'ssc_minmax_hw': [_P, _I, _I, _I, _I, _P, _P, _L, _P],
    'ssc_concat_parts': [_P, _P],
    'ssc_mru_gate_merge': [_P, _P, _P, _P, _P, _I, _L, _I, _P],
    'ssc_colsum': [_P, _I, _L, _I, _P, _I, _P, _L, _P],
    'ssc_strided_copy': [_P, _I, _P, _I, _L, _I, _I, _P],
    'ssc_pool2': [_P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    'ssc_cbn_act_backward': [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _I, _P, _L, _P],
    'ssc_prelu_backward': [_P, _I, _P, _P, _I, _L, _I, _P, _I, _I, _P, _I, _P, _L, _P],
    'ssc_minmax_gate_backward': [_P, _P, _P, _I, _I, _I, _P, _P, _L, _P],
    'ssc_mru_gate_merge_backward': [_P, _P, _P, _P, _P, _P, _I, _L, _I, _P],
    'ssc_mru_blend_backward': [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'ssc_mru_in2_gate_backward': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'ssc_mru_blend': [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'ssc_fc_small_fwd': [_P, _P, _P, _I, _I, _I, _P, _P],
    'ssc_fc_small_bwd': [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _P],
}


class CatPart(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ab', C.c_void_p), ('gate', C.c_void_p), ('mnmx', C.c_void_p),
                ('ld', C.c_int32), ('C', C.c_int32), ('ab_sample_stride', C.c_int32), ('act', C.c_int32),
                ('upsample', C.c_int32), ('_pad', C.c_int32)]


class CatDesc(C.Structure):
    _fields_ = [('p', CatPart * 3), ('out', C.c_void_p), ('nparts', C.c_int32), ('ldo', C.c_int32),
                ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('_pad', C.c_int32)]


def _declare(l):
    for name, args in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int


SK_ENABLED = os.environ.get('SSC_STREAMK', '1') != '0'      # in-kernel stream-K decomposition of the conv launches
LAUNCHES = 0        # bumped by every C-ABI call (lets the trainer tell an empty graph segment from a real one)


def check(rc, what):
    global LAUNCHES
    LAUNCHES += 1
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32, torch.float64, torch.uint8), (t.device, t.dtype)
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------
# workspace (split-K slabs, reduction partials) -- one per device, grown on demand
# ---------------------------------------------------------------------------
_ws = {}


def workspace(nbytes=256 << 20):
    """Split-K / reduction scratch of the CURRENT stream (one buffer per device and stream: kernels of the filter-
    gradient side stream must not share partial-sum slabs with the main stream)."""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
    w = _ws.get(key)
    if w is None or w.numel() * 4 < nbytes:
        w = torch.empty(nbytes // 4, dtype=torch.float32, device='cuda')
        _ws[key] = w
    return w


# Filter gradients on a side stream.  In a backward pass the data-gradient chain (dgrad -> norm backward -> dgrad ...)
# is the critical path; the filter gradient of each layer only consumes what that chain has already produced and its
# result is not needed before the optimizer.  When the trainer sets WGRAD_STREAM, conv_wgrad / deconv_wgrad launch
# there (after waiting for everything issued so far on the current stream), so their workgroups fill the CUs that the
# chain's small layers and kernel tails leave idle; join_wgrad() makes the current stream wait for them.
WGRAD_STREAM = None
WGRAD_SIDE_MAX_PIXELS = int(_dev_env('SSC_WGRAD_SIDE_PIXELS', '0')) or None   # None = every layer
_wgrad_pending = False


def join_wgrad():
    global _wgrad_pending
    if _wgrad_pending and WGRAD_STREAM is not None:
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)
    _wgrad_pending = False


class View(object):
    """Host-side mirror of ssc_gview: NHWC tensor(s) + folded norm + activation."""

    def __init__(self, s0, s1=None, ab0=None, act=ACT_NONE, ab1=None, act1=-1):
        assert s0.dim() == 4 and s0.is_contiguous()
        self.s0, self.s1, self.ab0, self.ab1, self.act, self.act1 = s0, s1, ab0, ab1, act, act1
        self.N, self.H, self.W, self.C0 = s0.shape
        self.C1 = 0
        if s1 is not None:
            assert s1.is_contiguous() and s1.shape[:3] == s0.shape[:3]
            self.C1 = s1.shape[3]
        self.C = self.C0 + self.C1
        assert ab0 is None or (ab0.numel() == 2 * self.C0 and ab0.is_contiguous())
        assert ab1 is None or (ab1.numel() == 2 * self.C1 and ab1.is_contiguous())

    def c(self):
        g = GView()
        g.s0 = self.s0.data_ptr()
        g.s1 = self.s1.data_ptr() if self.s1 is not None else None
        g.ab0 = self.ab0.data_ptr() if self.ab0 is not None else None
        g.ab1 = self.ab1.data_ptr() if self.ab1 is not None else None
        g.C0, g.C1, g.H, g.W, g.act, g.act1 = self.C0, self.C1, self.H, self.W, self.act, self.act1
        return g


# When PROFILE is a list, every implicit-GEMM launch is bracketed by HIP events on the launch
# stream and (kernel name, algorithmic FLOPs, start, stop, (M, N, K), algorithmic HBM bytes) is appended
# (bench.py roofline leg).
PROFILE = None


def _kernel_name(fn, d):
    buf = C.create_string_buffer(64)
    getattr(lib(), fn)(C.byref(d), buf, 64)
    return buf.value.decode()


SK_FLAG_WORDS = 8192
_SK_MAX_STREAMS = 32
_sk_flags = {}
_sk_pool = {}           # device -> [_SK_MAX_STREAMS, SK_FLAG_WORDS] int32: every stream's flag array is a row of it
_sk_tags = {}           # sk_tag -> copy of the descriptor of that launch
_sk_eager = []          # tags of launches issued eagerly, oldest first: only the last _SK_TAG_RING of them are remembered
_sk_next = [0]          # tags only grow (31 bits: the flag word is 0x80000000 | tag), so a tag frozen into a captured graph
_SK_TAG_RING = 1 << 14  # never comes to name a later launch; the descriptors of captured launches are kept for good
_sk_configured = False


class HandoffTimeout(RuntimeError):
    """An owner workgroup of a conv launch gave up waiting for a K slice: that launch stored a partial sum."""


def _sk_configure():
    """SSC_SK_TIMEOUT_MS (default 20000: a deadlock guard only), SSC_SK_TEST_WITHHOLD=1 (test hook: producers never raise
    their flags, so every sliced tile reaches the bound)."""
    global _sk_configured
    if not _sk_configured:
        _sk_configured = True
        ms, wh = int(os.environ.get('SSC_SK_TIMEOUT_MS', '0')), int(os.environ.get('SSC_SK_TEST_WITHHOLD', '0'))
        if ms > 0 or wh:
            check(lib().ssc_sk_configure(ms, wh), 'ssc_sk_configure')


def sk_flags():
    """Stream-K hand-off flags of the CURRENT stream (ssc_conv_desc.sk_flags): zero when created, left zero by every
    kernel that uses them; one array per device and stream because launches on different streams run concurrently."""
    dev = torch.cuda.current_device()
    key = (dev, torch.cuda.current_stream().cuda_stream)
    f = _sk_flags.get(key)
    if f is None:
        _sk_configure()
        pool = _sk_pool.get(dev)
        if pool is None:
            pool = _sk_pool[dev] = torch.zeros((_SK_MAX_STREAMS, SK_FLAG_WORDS), dtype=torch.int32, device='cuda')
        n = sum(1 for k in _sk_flags if k[0] == dev)
        # beyond the pool (never seen: a trainer uses five streams) a stream gets an array of its own
        f = pool[n] if n < _SK_MAX_STREAMS else torch.zeros(SK_FLAG_WORDS, dtype=torch.int32, device='cuda')
        _sk_flags[key] = f
    return f


def _sk_tag(d):
    """Tag a launch that may split tiles inside the kernel and remember its descriptor for the error message."""
    _sk_next[0] = (_sk_next[0] % 0x7ffffffe) + 1
    tag = _sk_next[0]
    d.sk_tag = tag
    _sk_tags[tag] = ConvDesc.from_buffer_copy(d)
    if not torch.cuda.is_current_stream_capturing():
        _sk_eager.append(tag)
        if len(_sk_eager) > _SK_TAG_RING:
            for old in _sk_eager[:_SK_TAG_RING // 2]:
                _sk_tags.pop(old, None)
            del _sk_eager[:_SK_TAG_RING // 2]


def sk_timeouts():
    """Number of flag arrays whose last word reports a hand-off that timed out (must be 0)."""
    n = 0
    for pool in _sk_pool.values():
        n += int((pool[:, SK_FLAG_WORDS - 1] != 0).sum().item())
    return n + sum(int(f[SK_FLAG_WORDS - 1].item() != 0) for f in _sk_flags.values() if f.dim() == 1 and f._base is None)


def check_sk(where=''):
    """Raise HandoffTimeout if any conv launch since the last check reported a hand-off timeout (igemm.hip: the owner of a
    sliced tile then stored a partial sum -- a silently wrong activation or gradient otherwise).  One small reduction and one
    host read: called wherever results are read back anyway (loss fetch, snapshot, end of a benchmark, smoke).  The flag
    arrays are zeroed before raising, so a caller that catches the error can go on (e.g. with SSC_STREAMK=0)."""
    bad = []
    for pool in _sk_pool.values():
        words = pool[:, SK_FLAG_WORDS - 1]
        if bool((words != 0).any().item()):
            bad += [int(w) & 0x7fffffff for w in words.cpu().tolist() if w != 0]
            torch.cuda.synchronize()        # launches still running on the other streams wait on flags of this pool
            pool.zero_()
    for f in _sk_flags.values():
        if f.dim() == 1 and f._base is None and int(f[SK_FLAG_WORDS - 1].item()) != 0:
            bad.append(int(f[SK_FLAG_WORDS - 1].item()) & 0x7fffffff)
            torch.cuda.synchronize()
            f.zero_()
    if not bad:
        return
    what = []
    for tag in bad:
        d = _sk_tags.get(tag)
        if d is None:
            what.append('launch tag %d' % tag)
        else:
            what.append('%s M=%d N=%d K=%d (launch tag %d)' % (_kernel_name('ssc_conv_forward_kernel_name', d),
                                                              d.NB * d.PH * d.PW * d.nphase, d.Nn, d.TH * d.TW * d.k_real, tag))
    raise HandoffTimeout('in-launch K-slice hand-off timed out%s: %s -- the outputs of these launches are partial sums; '
                         'set SSC_STREAMK=0 to run without in-launch combining' % (' (' + where + ')' if where else '',
                                                                                 '; '.join(what)))


_sk_ring = {'host': None, 'i': 0}


def check_sk_begin():
    """The asynchronous half of ``check_sk``: "did any launch so far report a hand-off timeout" is reduced on the device and
    copied to pinned host memory behind everything queued on the current stream; nothing waits.  ``check_sk_end(handle)``
    waits for THAT copy only (not for work queued later) and runs the full check -- names, zeroing, HandoffTimeout -- if it
    says yes.  For callers that read a step's results while the next step is already running (main_procedure.train)."""
    parts = [pool[:, SK_FLAG_WORDS - 1] for pool in _sk_pool.values()]
    parts += [f[SK_FLAG_WORDS - 1:] for f in _sk_flags.values() if f.dim() == 1 and f._base is None]
    if not parts:
        return None
    any_dev = (parts[0] if len(parts) == 1 else torch.cat([x.reshape(-1) for x in parts])).ne(0).any()
    if _sk_ring['host'] is None:
        _sk_ring['host'] = torch.zeros(64, dtype=torch.bool).pin_memory()
    _sk_ring['i'] = (_sk_ring['i'] + 1) % 64
    slot = _sk_ring['host'][_sk_ring['i']:_sk_ring['i'] + 1]
    slot.copy_(any_dev.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return slot, ev


def check_sk_end(handle, where=''):
    if handle is None:
        return
    slot, ev = handle
    ev.synchronize()
    if bool(slot[0]):
        check_sk(where)


# ---------------------------------------------------------------------------
# bf16-split filters (igemm_bf16.hip)
# ---------------------------------------------------------------------------
# SSC_ARITH=fp32: every contraction on the exact-fp32 MFMA.  Default: the layers whose channel counts allow it run as six bf16
# products per fp32 product with fp32 accumulation (fp32-grade results, 6/16 of the matrix cycles); their filters are split into
# three bf16 planes ahead of the launches -- lazily at first use, again whenever torch modifies the tensor (its version counter),
# and by refresh_splits() after every optimizer launch (which writes the weights behind torch's back).
ARITH_BF16 = os.environ.get('SSC_ARITH', 'bf16x6').lower() not in ('fp32', 'f32', 'float32')
_SPLITS = {}            # (data_ptr, taps, c0, c1, orient) -> _Split; the entry keeps the filter tensor alive (its address stays taken)
_SPLIT_TABLES = {}      # tuple of keys -> (device job table, total threads)
_PARAM_RANGES = []      # [lo, hi) byte ranges of the flat parameter buffers (ParamStore scopes): filters inside them are PARAMETERS
_VOLATILE_MAX = 256     # entries of filters that are not parameters (see filter_split)
_SPLIT_ALWAYS = _dev_env('SSC_SPLIT_ALWAYS', '0') == '1'      # diagnostic: every filter treated as volatile


class _Split(object):
    # gen: creation order of PARAMETER entries (a captured optimizer step refreshes the entries that existed at its capture;
    # refresh_new_splits covers the younger ones).  pinned: created or used while a stream was capturing -- a captured graph
    # holds its buffer's address, so the entry is never evicted.
    __slots__ = ('key', 'w', 'buf', 'kc', 'nbp', 'threads', 'version', 'taps', 'c0', 'c1', 'orient', 'param', 'gen', 'pinned')


_split_gen = [0]        # number of parameter entries ever created (see _Split.gen)


def split_generation():
    """The creation counter of parameter plane entries: a trainer notes it when it captures an optimizer step and hands it to
    refresh_new_splits() after every replay."""
    return _split_gen[0]


def register_param_buffer(flat):
    """A flat parameter buffer (ParamStore scope): filters that are views of it keep PERSISTENT bf16 planes, refreshed by
    refresh_splits() behind every optimizer launch.  Returns the range for release_param_buffer."""
    rng = (flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size())
    _PARAM_RANGES.append(rng)
    return rng


def release_param_buffer(rng):
    """Forget a parameter buffer and the planes of its filters (called when its scope is collected)."""
    if rng in _PARAM_RANGES:
        _PARAM_RANGES.remove(rng)
    for k in [k for k in _SPLITS if rng[0] <= k[0] < rng[1]]:
        del _SPLITS[k]
    for tk in [tk for tk in _SPLIT_TABLES if any(rng[0] <= k[0] < rng[1] for k in tk)]:
        del _SPLIT_TABLES[tk]


def _split_launch(e):
    check(lib().ssc_filter_split(ptr(e.w), e.taps, e.c0, e.c1, e.orient, ptr(e.buf), stream_ptr()), 'ssc_filter_split')


def filter_split(w, orient):
    """The bf16 planes of filter w [KH,KW,c0,c1] in orientation ``orient`` (0: k = c0, n = c1; 1: k = c1, n = c0).
    * A PARAMETER (a view of a registered flat buffer) keeps its planes: split at first use, again when torch modifies the
      tensor (version counter), and by refresh_splits() behind the optimizer launches, which torch does not see.
    * Any other filter -- spectral-normed weights, transposed copies, whatever a kernel writes each forward pass; a test's
      tensor -- is VOLATILE: nobody can tell when its contents change, so it is split again in front of EVERY launch, on the
      launch's stream."""
    KH, KW, c0, c1 = w.shape
    key = (w.data_ptr(), KH * KW, c0, c1, orient)
    e = _SPLITS.get(key)
    if e is None:
        if not _SPLITS:
            check(lib().ssc_bf16_prepare(), 'ssc_bf16_prepare')     # per-device constants, outside any capture
        kc, nbp, nbytes, threads = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int64(0)
        check(lib().ssc_filter_split_geom(KH * KW, c0, c1, orient, C.byref(kc), C.byref(nbp), C.byref(nbytes), C.byref(threads)),
              'ssc_filter_split_geom')
        e = _Split()
        e.key, e.w, e.kc, e.nbp, e.threads, e.version = key, w, kc.value, nbp.value, threads.value, None
        e.taps, e.c0, e.c1, e.orient = KH * KW, c0, c1, orient
        e.param = (not _SPLIT_ALWAYS) and any(lo <= key[0] < hi for lo, hi in _PARAM_RANGES)
        e.buf = torch.empty(nbytes.value, dtype=torch.uint8, device=w.device)
        e.pinned, e.gen = False, 0
        if e.param:
            _split_gen[0] += 1
            e.gen = _split_gen[0]
        else:
            # evict the oldest volatile entries nobody can still name: never one that a captured graph launches into (pinned)
            vol = [k for k, v in _SPLITS.items() if not v.param and not v.pinned]
            if len(vol) >= _VOLATILE_MAX and not torch.cuda.is_current_stream_capturing():
                torch.cuda.synchronize()        # nothing in flight reads the planes about to be dropped
                for k in vol[:_VOLATILE_MAX // 2]:
                    del _SPLITS[k]
        _SPLITS[key] = e
    if torch.cuda.is_current_stream_capturing():
        e.pinned = True
    if not e.param:
        _split_launch(e)
        return e
    if e.version != w._version:
        # A lazy split happens on whatever stream meets the filter first, while the trainers read one filter from several
        # streams (run-ahead forward, caption stream, the two discriminator passes): outside a capture the split is therefore
        # made visible to EVERY stream before anybody can use the entry -- the device drains in front of it (nobody still
        # reads the old planes) and behind it.  This is a first-use / weights-replaced-through-torch event, not a per-step one
        # (the optimizer path is refresh_splits, ordered by the stream that ran the optimizer).
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            torch.cuda.synchronize()
        _split_launch(e)
        if not capturing:
            torch.cuda.synchronize()
        e.version = w._version
    return e


def resplit_stale():
    """Split again every parameter filter whose tensor torch has modified since its planes were made (ParamStore.load_dict /
    load_state_dict / initialize, any copy_ into a view of the flat buffer).  filter_split() notices that when Python meets the
    filter; a REPLAYED graph never comes by there -- its launches hold the planes' addresses -- so the trainers call this in
    front of every replay (a loop over some dozens of integers when nothing changed).  Returns the number of filters split."""
    stale = [e for e in _SPLITS.values() if e.param and e.version is not None and e.version != e.w._version]
    if not stale:
        return 0
    assert not torch.cuda.is_current_stream_capturing(), 'weights were replaced through torch during a capture'
    torch.cuda.synchronize()        # nobody still reads the old planes (any stream)
    for e in stale:
        _split_launch(e)
        e.version = e.w._version
    torch.cuda.synchronize()        # ... and every stream sees the new ones
    return len(stale)


def refresh_new_splits(flat, since):
    """refresh_splits() for the parameter entries of ``flat`` created after generation ``since`` (split_generation() at the capture
    of an optimizer step): a captured step refreshes only the entries that existed when it was captured, so a filter that first
    met a bf16 launch later (inference at another batch size or image size, another orientation) would keep the planes of the
    weights of that moment.  Called behind every replay of a captured optimizer step; nothing to do in the usual case."""
    if _split_gen[0] <= since:
        return 0
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
    es = [e for e in _SPLITS.values() if e.param and e.gen > since and lo <= e.key[0] < hi]
    if es:
        _refresh_entries(es)
    return len(es)


def refresh_splits(flat=None):
    """Split again every registered filter that lives inside the flat parameter buffer ``flat`` (all of them when None): one
    batched launch.  Called behind every optimizer launch (the kernel writes the weights without torch noticing)."""
    if not _SPLITS:
        return
    if flat is not None:
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        es = [e for e in _SPLITS.values() if e.param and lo <= e.key[0] < hi]
    else:
        es = [e for e in _SPLITS.values() if e.param]
    if not es:
        return
    _refresh_entries(es)


def _refresh_entries(es):
    if torch.cuda.is_current_stream_capturing():
        for e in es:
            e.pinned = True
    tk = tuple(e.key for e in es)
    tab = _SPLIT_TABLES.get(tk)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():     # no host-to-device copy inside a capture: one launch per filter
            for e in es:
                _split_launch(e)
            return
        jobs = (SplitJob * len(es))()
        first = 0
        for j, e in zip(jobs, es):
            j.w, j.dst, j.taps, j.c0, j.c1, j.orient, j.first_thread = e.w.data_ptr(), e.buf.data_ptr(), e.taps, e.c0, e.c1, e.orient, first
            first += e.threads
        host = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        tab = (host.to(es[0].w.device), first)
        _SPLIT_TABLES[tk] = tab
    check(lib().ssc_filter_split_batch(ptr(tab[0]), len(es), tab[1], stream_ptr()), 'ssc_filter_split_batch')


_BF_PARTIAL = _dev_env('SSC_BF_PARTIAL', '1') == '1'      # A/B: the partial-chunk form of the bf16 conv kernel


def bf_form(d):
    """Which bf16-split form of the conv kernel a launch descriptor qualifies for (the library decides again: fwd_is_bf):
    'uniform' -- every 32-wide K-tile inside one tap and one source; 'partial' -- ONE source with any multiple of 4 channels
    above 32 (MRU's materialised concats: the kernel masks the last chunk of a tap); None -- the exact-fp32 kernels."""
    if (d.n_off % 32) or d.Nstore <= 32:
        return None
    if d.NB * d.PH * d.PW * d.nphase < 64:        # a handful of rows: nothing to win
        return None
    if not ((d.x.C0 % 32) or (d.x.C1 % 32) or d.k_real != d.x.C0 + d.x.C1):
        return 'uniform'
    if _BF_PARTIAL and d.x.C1 == 0 and d.x.C0 > 32 and d.x.C0 % 4 == 0 and d.x.C0 % 32 != 0 and \
            (d.wC1 if d.bmode else d.wC0) == d.k_real and -(-d.k_real // 32) == -(-d.x.C0 // 32):
        return 'partial'
    return None


def _attach_split(d, w):
    """Give the launch its filter's bf16 planes when it can run on the bf16 pipe."""
    if not ARITH_BF16 or w is None or bf_form(d) is None:
        return
    e = filter_split(w, 1 if d.bmode else 0)
    d.wsplit, d.ws_kc, d.ws_nbp = e.buf.data_ptr(), e.kc, e.nbp


class BnBwdSums(object):
    """Partial sums of a norm backward gathered from the epilogues of the launches that produce its incoming gradients
    (ssc_conv_forward_bnbwd).  x2d [M, C]: the normed tensor (raw), ab / stats its folded norm.  ``take(act)`` gives the
    argument for conv_dgrad / deconv_dgrad(..., bnbwd=...); bn_act_backward(..., pre=this) uses the rows when every
    gradient source delivered them and falls back to its own pass otherwise."""

    def __init__(self, x2d, ab, stats, buf):
        self.x2d, self.ab, self.stats, self.buf = x2d, ab, stats, buf
        self.rows, self.sources, self.missed = 0, 0, 0

    @staticmethod
    def rows_needed(M, sources=1):
        """Rows of ``buf`` that always suffice: one per 64-row tile and sub-pixel phase, per gradient source."""
        return sources * 4 * ((M // 4 + 63) // 64 + 1)

    def take(self, act):
        return (self, act)


# Set by a trainer around the steps whose chains run side by side on several streams (ssc_conv_desc.lds_hint): the conv launches
# then take the form with the smaller LDS footprint, so that workgroups of different chains fit one CU together.
CO_RUN = False


def _run_conv(d, bn=None, bnbwd=None, minmax=None):
    """bn = (scale, offset, ab, stats[, eps]): also fold the batch-statistics norm of the conv's output (the whole
    [rows, ldc] output must be the normed tensor).  bnbwd = BnBwdSums.take(act): the output is a gradient w.r.t. that
    activated norm; its backward sums come out of the epilogue when the launch qualifies."""
    ws = workspace()
    d.sk_flags = sk_flags().data_ptr() if SK_ENABLED else None
    d.lds_hint = 1 if CO_RUN else 0
    if SK_ENABLED:
        _sk_tag(d)

    def launch():
        if isinstance(bnbwd, list):
            # two normed tensors side by side: the columns [0, C0) are the gradient w.r.t. the first, the rest w.r.t. the second
            (s0, a0), (s1, a1) = bnbwd
            C0, C1 = s0.x2d.shape[1], s1.x2d.shape[1]
            assert d.Nstore == d.ldc == C0 + C1, (d.Nstore, d.ldc, C0, C1)
            sites = []
            for sm, act in ((s0, a0), (s1, a1)):
                assert sm.x2d.shape[0] == d.NB * d.OH * d.OW and sm.ab.numel() == 2 * sm.x2d.shape[1]
                part = sm.buf[sm.rows:]
                st = BnBwdSite()
                st.x, st.ab, st.stats, st.partial = sm.x2d.data_ptr(), sm.ab.data_ptr(), sm.stats.data_ptr(), part.data_ptr()
                st.partial_bytes, st.ldx, st.act = part.numel() * 4, sm.x2d.stride(0), act
                sites.append(st)
            n = C.c_int(0)
            check(lib().ssc_conv_forward_bnbwd2(C.byref(d), ptr(ws), ws.numel() * 4, C.byref(sites[0]), C.byref(sites[1]), C0,
                                                C.byref(n), stream_ptr()), 'ssc_conv_forward_bnbwd2')
            for sm in (s0, s1):
                sm.sources += 1
                if n.value > 0:
                    sm.rows += n.value
                else:
                    sm.missed += 1
        elif bnbwd is not None:
            sums, act = bnbwd
            C2 = 2 * sums.x2d.shape[1]
            # the epilogue indexes the normed tensor and its tables with the conv's own column count and pixel rows
            assert d.Nstore == d.ldc == sums.x2d.shape[1], (d.Nstore, d.ldc, sums.x2d.shape)
            assert sums.ab.numel() == 2 * d.Nstore and sums.stats.numel() == 2 * d.Nstore
            assert sums.x2d.shape[0] == d.NB * d.OH * d.OW, (sums.x2d.shape, d.NB, d.OH, d.OW)
            part = sums.buf[sums.rows:]
            n = C.c_int(0)
            check(lib().ssc_conv_forward_bnbwd(C.byref(d), ptr(ws), ws.numel() * 4, ptr(sums.x2d), sums.x2d.stride(0),
                                               ptr(sums.ab), ptr(sums.stats), act, ptr(part), part.numel() * 4,
                                               C.byref(n), stream_ptr()), 'ssc_conv_forward_bnbwd')
            sums.sources += 1
            if n.value > 0:
                assert part.shape[1] == C2
                sums.rows += n.value
            else:
                sums.missed += 1
        elif minmax is not None:        # mnmx [N,2,C] <- per-sample extrema of the (activated) output, out of the epilogue
            assert minmax.shape == (d.NB, 2, d.Nstore) and minmax.is_contiguous(), (minmax.shape, d.NB, d.Nstore)
            check(lib().ssc_conv_forward_minmax(C.byref(d), ptr(ws), ws.numel() * 4, ptr(minmax), stream_ptr()),
                  'ssc_conv_forward_minmax')
        elif bn is None:
            check(lib().ssc_conv_forward(C.byref(d), ptr(ws), ws.numel() * 4, stream_ptr()), 'ssc_conv_forward')
        else:
            eps = bn[4] if len(bn) > 4 else 1e-5
            check(lib().ssc_conv_forward_bn(C.byref(d), ptr(ws), ws.numel() * 4, ptr(bn[0]), ptr(bn[1]), eps, ptr(bn[2]),
                                            ptr(bn[3]), stream_ptr()), 'ssc_conv_forward_bn')

    if PROFILE is None:
        launch()
        return
    flops = 2.0 * d.NB * d.PH * d.PW * d.nphase * d.TH * d.TW * d.k_real * d.Nn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    # algorithmic HBM bytes: every input element, filter element and output element once
    nbytes = 4.0 * (d.NB * d.x.H * d.x.W * (d.x.C0 + d.x.C1) + d.nphase * d.TH * d.TW * d.k_real * d.Nn +
                    d.NB * d.PH * d.PW * d.nphase * d.Nstore * (2 if d.accumulate else 1))
    PROFILE.append((_kernel_name('ssc_conv_forward_kernel_name', d), flops, e0, e1,
                    (d.NB * d.PH * d.PW * d.nphase, d.Nn, d.TH * d.TW * d.k_real), nbytes))


def _run_wgrad(d, side=False):
    global _wgrad_pending
    d.exact = 0 if ARITH_BF16 else 1
    if side and WGRAD_STREAM is not None and PROFILE is None and \
            (WGRAD_SIDE_MAX_PIXELS is None or d.NB * d.PH * d.PW <= WGRAD_SIDE_MAX_PIXELS):
        WGRAD_STREAM.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(WGRAD_STREAM):
            ws = workspace()
            check(lib().ssc_conv_wgrad(C.byref(d), ptr(ws), ws.numel() * 4, stream_ptr()), 'ssc_conv_wgrad')
        _wgrad_pending = True
        return
    ws = workspace()
    if PROFILE is None:
        check(lib().ssc_conv_wgrad(C.byref(d), ptr(ws), ws.numel() * 4, stream_ptr()), 'ssc_conv_wgrad')
        return
    flops = 2.0 * d.NB * d.PH * d.PW * d.TH * d.TW * d.Cg_real * d.Nn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib().ssc_conv_wgrad(C.byref(d), ptr(ws), ws.numel() * 4, stream_ptr()), 'ssc_conv_wgrad')
    e1.record()
    nbytes = 4.0 * (d.NB * d.g.H * d.g.W * (d.g.C0 + d.g.C1) + d.NB * d.PH * d.PW * (d.d.C0 + d.d.C1) +
                    d.TH * d.TW * d.Cg_real * d.Nn * (2 if d.accumulate else 1))
    PROFILE.append((_kernel_name('ssc_conv_wgrad_kernel_name', d), flops, e0, e1,
                    (d.TH * d.TW * d.Cg_real, d.Nn, d.NB * d.PH * d.PW), nbytes))


def _out_geom(out, coff):
    assert out.dim() == 4 and out.is_contiguous()
    return out.data_ptr() + 4 * coff, out.shape[1], out.shape[2], out.shape[3]


def same_pad_before(size, k, stride):
    """TF SAME padding rule: pad_before = total // 2, the extra element goes after (bottom/right)."""
    out = -(-size // stride)
    return max((out - 1) * stride + k - size, 0) // 2


def conv_forward(x, w, stride, pad, out, coff=0, nstore=None, bias=None, epi=0, accumulate=False, same=False, bn=None,
                 w_nk=None, minmax=None):
    """tf.pad + tf.nn.conv2d(VALID): x View, w [KH,KW,Cin_real,Cout] -> out[..., coff:coff+Cout].
    same=True: tf.nn.conv2d(padding='SAME') -- output ceil(in/stride), asymmetric pad (mru.py:125, conv_ex).
    w_nk: the same filter as [KH,KW,Cout,Cin] (a transposed copy, e.g. ``transpose_filter``): the launch then reads the [n][k]
    orientation, which the few-output kernels take through scalar loads."""
    KH, KW, ci, co = w.shape
    d = ConvDesc()
    d.x = x.c()
    d.w, d.bias = w.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.out, OH, OW, ldc = _out_geom(out, coff)
    if same:
        d.NB, d.PH, d.PW = x.N, -(-x.H // stride), -(-x.W // stride)
        pad = same_pad_before(x.H, KH, stride)
        assert pad == same_pad_before(x.W, KW, stride)
    else:
        d.NB, d.PH, d.PW = x.N, (x.H + 2 * pad - KH) // stride + 1, (x.W + 2 * pad - KW) // stride + 1
    assert (OH, OW) == (d.PH, d.PW), ((OH, OW), (d.PH, d.PW))
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x, d.nphase = KH, KW, stride, -pad, -pad, 1
    d.ky0, d.kx0, d.kstep = 0, 0, 1
    d.KH, d.KW, d.wC0, d.wC1, d.bmode, d.k_real = KH, KW, ci, co, 0, ci
    if w_nk is not None:
        assert tuple(w_nk.shape) == (KH, KW, co, ci) and w_nk.is_contiguous()
        d.w, d.wC0, d.wC1, d.bmode = w_nk.data_ptr(), co, ci, 1
    assert ci <= x.C
    d.n_off, d.Nn, d.Nstore = 0, co, (nstore if nstore is not None else co)
    d.OH, d.OW, d.ldc, d.out_stride, d.ooff_y, d.ooff_x = OH, OW, ldc, 1, 0, 0
    d.epi, d.accumulate = epi, int(accumulate)
    assert minmax is None or (bn is None and coff == 0)
    _attach_split(d, w_nk if w_nk is not None else w)
    _run_conv(d, _bn_arg(bn, d, coff, out), minmax=minmax)


def transpose_filter(w, out):
    """out [KH,KW,Cout,Cin] <- w [KH,KW,Cin,Cout] (one small launch; for conv_forward(..., w_nk=out))."""
    KH, KW, ci, co = w.shape
    assert tuple(out.shape) == (KH, KW, co, ci) and w.is_contiguous() and out.is_contiguous()
    nhwc_to_nchw(w.view(KH * KW, 1, ci, co), out.view(KH * KW, co, 1, ci))
    return out


def _bn_arg(bn, d, coff, out):
    """The fused conv + norm entry covers the plain case (the conv fills whole rows of ``out``); anything else keeps the two
    calls apart."""
    if bn is None:
        return None
    assert coff == 0 and d.Nstore == out.shape[3] == bn[0].numel() and bn[2].numel() == 2 * d.Nstore, \
        (coff, d.Nstore, out.shape, bn[0].shape)
    return bn


def deconv_forward(x, f, out, coff=0, nstore=None, epi=0, bn=None, _desc_only=False):
    """tf.nn.conv2d_transpose(k=4, s=2, SAME): x View [N,H,W,Cin], f [4,4,Cout,Cin] -> out [N,2H,2W,*]."""
    KH, KW, co, ci = f.shape
    assert KH == 4 and KW == 4 and ci <= x.C        # ci < x.C: 3-channel tensors padded to 4 (BG region branch)
    d = ConvDesc()
    d.x = x.c()
    d.w, d.bias = f.data_ptr(), None
    d.out, OH, OW, ldc = _out_geom(out, coff)
    assert (OH, OW) == (2 * x.H, 2 * x.W)
    d.NB, d.PH, d.PW = x.N, x.H, x.W
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x, d.nphase = 2, 2, 1, 0, 0, 4
    d.ky0, d.kx0, d.kstep = 0, 0, -2
    d.KH, d.KW, d.wC0, d.wC1, d.bmode, d.k_real = 4, 4, co, ci, 1, ci
    d.n_off, d.Nn, d.Nstore = 0, co, (nstore if nstore is not None else co)
    d.OH, d.OW, d.ldc, d.out_stride, d.ooff_y, d.ooff_x = OH, OW, ldc, 2, 0, 0
    d.epi, d.accumulate = epi, 0
    if _desc_only:
        return d
    _attach_split(d, f)
    _run_conv(d, _bn_arg(bn, d, coff, out))


def conv_dgrad(dy, w, stride, pad, out, n_off=0, nn=None, k_real=None, nstore=None, accumulate=False, bnbwd=None, coff=0,
               _desc_only=False):
    """Gradient of a conv w.r.t. its input channels [n_off, n_off+nn): dy View -> out[..., coff:coff+nstore] [N,Hin,Win,*].
    stride 2: the k=4 pad-1 conv (4 sub-pixel phases).  stride 1: any square kernel, ``pad`` = padding before
    (SAME: (k-1)//2, the extra element after), input size taken from ``out``."""
    KH, KW, ci, co = w.shape
    nn = ci - n_off if nn is None else nn
    d = ConvDesc()
    d.x = dy.c()
    d.w, d.bias = w.data_ptr(), None
    d.out, OH, OW, ldc = _out_geom(out, coff)
    d.NB = dy.N
    if stride == 2:
        assert KH == 4 and KW == 4 and pad == 1 and (OH, OW) == (2 * dy.H, 2 * dy.W)
        d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase, d.kstep, d.out_stride = dy.H, dy.W, 2, 2, 1, 4, -2, 2
        d.ky0 = d.kx0 = 0
        d.ioff_y = d.ioff_x = 0
    else:
        assert stride == 1 and KH == KW and abs(OH + 2 * pad - KH + 1 - dy.H) <= 1, (OH, dy.H, KH, pad)
        d.PH, d.PW, d.TH, d.TW, d.in_stride, d.nphase, d.kstep, d.out_stride = OH, OW, KH, KW, 1, 1, -1, 1
        d.ky0 = d.kx0 = KH - 1
        d.ioff_y = d.ioff_x = pad - (KH - 1)
    d.KH, d.KW, d.wC0, d.wC1, d.bmode = KH, KW, ci, co, 1
    d.k_real = co if k_real is None else k_real
    d.n_off, d.Nn, d.Nstore = n_off, nn, (nstore if nstore is not None else nn)
    d.OH, d.OW, d.ldc, d.ooff_y, d.ooff_x = OH, OW, ldc, 0, 0
    d.epi, d.accumulate = 0, int(accumulate)
    if _desc_only:
        return d
    _attach_split(d, w)
    _run_conv(d, bnbwd=bnbwd)


def head1_dgrad_bn_backward(dy, w, pad, x4d, ab, stats, act, dx4d, dscale=None, doffset=None, rowb=None):
    """Data gradient of a one-output conv head (w [k,k,512,1], stride 1) fused with the backward of the batch norm + activation
    of its input x4d [N,H,W,512]: dx4d <- d loss / d x4d, the gradient w.r.t. act(norm(x)) is never stored (head1.hip).
    rowb = (v [N,512], scale): a per-image term added to that gradient.  Returns False (nothing launched) when the shape is
    not the head's: callers then run conv_dgrad + bn_act_backward."""
    if tuple(w.shape[2:]) != (512, 1) or x4d.shape[-1] != 512 or ab is None:
        return False
    d = conv_dgrad(dy, w, 1, pad, dx4d, k_real=1, _desc_only=True)
    if not lib().ssc_head1_dgrad_supported(C.byref(d)):
        return False
    ws = workspace()
    check(lib().ssc_head1_dgrad_bn_backward(C.byref(d), ptr(x4d), ptr(ab), ptr(stats), act,
                                            ptr(rowb[0]) if rowb else None, float(rowb[1]) if rowb else 0.0, ptr(dx4d),
                                            ptr(dscale), ptr(doffset), ptr(ws), ws.numel() * 4, stream_ptr()),
          'ssc_head1_dgrad_bn_backward')
    return True


def deconv_dgrad(dy, f, out, n_off=0, nn=None, accumulate=False, bnbwd=None):
    """Gradient of the k=4 s=2 transposed conv w.r.t. its input channels: a stride-2 conv of dy with f as HWIO."""
    KH, KW, co, ci = f.shape
    nn = ci - n_off if nn is None else nn
    d = ConvDesc()
    d.x = dy.c()
    d.w, d.bias = f.data_ptr(), None
    d.out, OH, OW, ldc = _out_geom(out, 0)
    d.NB, d.PH, d.PW = dy.N, dy.H // 2, dy.W // 2
    assert (OH, OW) == (d.PH, d.PW) and co <= dy.C
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x, d.nphase = 4, 4, 2, -1, -1, 1
    d.ky0, d.kx0, d.kstep = 0, 0, 1
    d.KH, d.KW, d.wC0, d.wC1, d.bmode, d.k_real = 4, 4, co, ci, 0, co
    d.n_off, d.Nn, d.Nstore = n_off, nn, nn
    d.OH, d.OW, d.ldc, d.out_stride, d.ooff_y, d.ooff_x = OH, OW, ldc, 1, 0, 0
    d.epi, d.accumulate = 0, int(accumulate)
    _attach_split(d, f)
    _run_conv(d, bnbwd=bnbwd)


def conv_wgrad(x, dy, w_grad, stride, pad, accumulate=False):
    """dW[kh,kw,ci,co] = sum_pix x[pix@tap][ci] * dy[pix][co]  (w_grad in the conv's TF layout)."""
    KH, KW, ci, co = w_grad.shape
    d = WgradDesc()
    d.g, d.d = x.c(), dy.c()
    d.out = w_grad.data_ptr()
    d.NB, d.PH, d.PW = dy.N, dy.H, dy.W
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = KH, KW, stride, -pad, -pad
    d.Cg_real, d.Nn, d.ldc, d.accumulate = ci, co, co, int(accumulate)
    assert ci <= x.C and co <= dy.C
    _run_wgrad(d, side=True)


def deconv_wgrad(x, dy, f_grad, accumulate=False):
    """dF[kh,kw,co,ci] = sum_pix dy[pix@tap][co] * x[pix][ci]  (f_grad in the transposed-conv TF layout)."""
    KH, KW, co, ci = f_grad.shape
    d = WgradDesc()
    d.g, d.d = dy.c(), x.c()
    d.out = f_grad.data_ptr()
    d.NB, d.PH, d.PW = x.N, x.H, x.W
    d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = 4, 4, 2, -1, -1
    d.Cg_real, d.Nn, d.ldc, d.accumulate = co, ci, ci, int(accumulate)
    assert co <= dy.C and ci <= x.C      # ci < x.C: 3-channel tensors padded to 4 (BG region branch)
    _run_wgrad(d, side=True)


def _mat_view(a, ab=None, act=ACT_NONE):
    assert a.dim() == 2 and a.is_contiguous()
    return View(a.view(1, 1, a.shape[0], a.shape[1]), None, ab, act)


def matmul(a, b, out, bias=None, accumulate=False, a_ab=None, a_act=ACT_NONE):
    """out[M,N] = a[M,K] @ b[K,N] (+bias)   -- tf.matmul."""
    M, K = a.shape
    K2, N = b.shape
    assert K == K2 and out.shape == (M, N) and out.is_contiguous()
    conv_forward(_mat_view(a, a_ab, a_act), b.view(1, 1, K, N), 1, 0, out.view(1, 1, M, N), bias=bias,
                 accumulate=accumulate)


def matmul_nt(a, b, out, accumulate=False):
    """out[M,N] = a[M,K] @ b[N,K]^T."""
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and out.shape == (M, N)
    d = ConvDesc()
    d.x = _mat_view(a).c()
    d.w, d.bias, d.out = b.data_ptr(), None, out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x, d.nphase = 1, 1, M, 1, 1, 1, 0, 0, 1
    d.ky0, d.kx0, d.kstep, d.KH, d.KW, d.wC0, d.wC1, d.bmode, d.k_real = 0, 0, 1, 1, 1, N, K, 1, K
    d.n_off, d.Nn, d.Nstore, d.OH, d.OW, d.ldc, d.out_stride, d.ooff_y, d.ooff_x = 0, N, N, 1, M, N, 1, 0, 0
    d.epi, d.accumulate = 0, int(accumulate)
    _attach_split(d, b.view(1, 1, N, K))
    _run_conv(d)


def matmul_tn(a, b, out, accumulate=False, a_ab=None, a_act=ACT_NONE):
    """out[K,N] = a[M,K]^T @ b[M,N]."""
    M, K = a.shape
    M2, N = b.shape
    assert M == M2 and out.shape == (K, N) and out.is_contiguous()
    d = WgradDesc()
    d.g, d.d = _mat_view(a, a_ab, a_act).c(), _mat_view(b).c()
    d.out = out.data_ptr()
    d.NB, d.PH, d.PW, d.TH, d.TW, d.in_stride, d.ioff_y, d.ioff_x = 1, 1, M, 1, 1, 1, 0, 0
    d.Cg_real, d.Nn, d.ldc, d.accumulate = K, N, N, int(accumulate)
    _run_wgrad(d)


# ---------------------------------------------------------------------------
# layout + norm helpers
# ---------------------------------------------------------------------------
def nchw_to_nhwc(src, dst, coff=0):
    n, c, h, w = src.shape
    assert dst.shape[:3] == (n, h, w) and src.is_contiguous() and dst.is_contiguous()
    check(lib().ssc_nchw_to_nhwc(ptr(src), ptr(dst), n, c, h * w, dst.shape[3], coff, stream_ptr()), 'nchw_to_nhwc')


def nhwc_to_nchw(src, dst, coff=0):
    n, c, h, w = dst.shape
    assert src.shape[:3] == (n, h, w) and src.is_contiguous() and dst.is_contiguous()
    check(lib().ssc_nhwc_to_nchw(ptr(src), ptr(dst), n, c, h * w, src.shape[3], coff, stream_ptr()), 'nhwc_to_nchw')


def sketch_preprocess_u8(src_u8, thicken=False, out=None):
    """uint8 [N,H,W,3] (device) -> float [N,H,W,4] network input in [-1,1] (channel 3 zero), optionally thickened."""
    n, h, w, c = src_u8.shape
    assert c == 3 and src_u8.dtype == torch.uint8 and src_u8.is_contiguous()
    if out is None:
        out = torch.empty((n, h, w, 4), dtype=torch.float32, device=src_u8.device)
    check(lib().ssc_sketch_preprocess_u8(ptr(src_u8), n, h, w, int(bool(thicken)), ptr(out), stream_ptr()),
          'sketch_preprocess_u8')
    return out


def image_postprocess_u8(src_nhwc, coff=0, out=None):
    """float NHWC [N,H,W,ldc] with the tanh image in channels [coff, coff+3) -> uint8 [N,H,W,3] (truncating cast)."""
    n, h, w, ldc = src_nhwc.shape
    assert src_nhwc.is_contiguous() and coff + 3 <= ldc
    if out is None:
        out = torch.empty((n, h, w, 3), dtype=torch.uint8, device=src_nhwc.device)
    check(lib().ssc_image_postprocess_u8(ptr(src_nhwc), ldc, coff, n * h * w, ptr(out), stream_ptr()),
          'image_postprocess_u8')
    return out


def resample_u8(src_u8, new_h, new_w, coeffs_h, coeffs_v, chan=-1, out_hw=None, top=0, left=0, fill=255, out_channels=None):
    """PIL.Image.resize of a uint8 [H,W,C] device tensor (Pillow's 8-bit two-pass resampler, bit for bit).  coeffs_* =
    (bounds int32 [new,2], coefficients int32 [new,ks]) device tensors from input_pipeline.resample_coeffs, or None when that
    axis keeps its size.  chan >= 0: that channel only, replicated over ``out_channels``.  The result lands at (top, left) of a
    [out_hw[0], out_hw[1], channels] canvas filled with ``fill``."""
    H, W, Cc = src_u8.shape
    assert src_u8.dtype == torch.uint8 and src_u8.is_contiguous() and src_u8.is_cuda
    OC = out_channels if out_channels is not None else (Cc if chan < 0 else 1)
    OH, OW = out_hw if out_hw is not None else (new_h, new_w)
    tmp = torch.empty((H, new_w, 1 if chan >= 0 else Cc), dtype=torch.uint8, device=src_u8.device)
    dst = torch.empty((OH, OW, OC), dtype=torch.uint8, device=src_u8.device)
    bh, kh = coeffs_h if coeffs_h is not None else (None, None)
    bv, kv = coeffs_v if coeffs_v is not None else (None, None)
    check(lib().ssc_resample_u8(ptr(src_u8), H, W, Cc, chan, ptr(bh), ptr(kh), kh.shape[1] if kh is not None else 0, new_w,
                                ptr(bv), ptr(kv), kv.shape[1] if kv is not None else 0, new_h, ptr(tmp), ptr(dst), OH, OW, OC,
                                top, left, fill, stream_ptr()), 'ssc_resample_u8')
    return dst


def distance_map_u8(sk_u8):
    """--distance_map 1: uint8 sketches [N,R,R,3] (device) -> float [N,R,R,3], exact Euclidean distance to the nearest
    stroke voxel scaled to [0, 255] (input_pipeline.py:86-96)."""
    n, r, r2, c = sk_u8.shape
    assert r == r2 and c == 3 and sk_u8.dtype == torch.uint8 and sk_u8.is_contiguous()
    out = torch.empty((n, r, r, 3), dtype=torch.float32, device=sk_u8.device)
    ws = torch.empty(2 * out.numel() + n, dtype=torch.int32, device=sk_u8.device)
    check(lib().ssc_distance_map_u8(ptr(sk_u8), n, r, ptr(out), ptr(ws), ws.numel() * 4, stream_ptr()), 'distance_map_u8')
    return out


def decode_paired_u8(img_u8, sk_u8, size, noise=None, img_out=None, sk_out=None, distance_map=False):
    """Raw record images uint8 [N,R,R,3] (device) -> (image, sketch) float NCHW [N,3,size,size] in [-1,1], as
    input_pipeline.decode_paired_example does on the host.  noise: uniform [0,1/256) [N,size,size,3] or None.
    distance_map: the sketch is first replaced by its distance map (``distance_map_u8``)."""
    n, r, r2, c = img_u8.shape
    assert r == r2 and c == 3 and sk_u8.shape == img_u8.shape and img_u8.is_contiguous() and sk_u8.is_contiguous()
    assert img_u8.dtype == torch.uint8 and sk_u8.dtype == torch.uint8 and r % size == 0
    if img_out is None:
        img_out = torch.empty((n, 3, size, size), dtype=torch.float32, device=img_u8.device)
    if sk_out is None:
        sk_out = torch.empty((n, 3, size, size), dtype=torch.float32, device=img_u8.device)
    mnmx = torch.empty((n, 2), dtype=torch.float32, device=img_u8.device)
    skf = distance_map_u8(sk_u8) if distance_map else None
    check(lib().ssc_decode_paired_u8(ptr(img_u8), ptr(sk_u8), ptr(skf), n, r, size, ptr(noise), ptr(img_out),
                                     ptr(sk_out), ptr(mnmx), stream_ptr()), 'decode_paired_u8')
    return img_out, sk_out


def new_graph():
    """A hipGraph object for a step capture.  SSC_KEEP_GRAPHS=1 keeps the captured graph beside its executable so that
    graph_kernel_nodes() can count what a replay launches (bench.py reports it)."""
    return torch.cuda.CUDAGraph(keep_graph=True) if os.environ.get('SSC_KEEP_GRAPHS') == '1' else torch.cuda.CUDAGraph()


def graph_kernel_nodes(g):
    """Number of kernel nodes of a captured torch.cuda.CUDAGraph (needs SSC_KEEP_GRAPHS=1 at capture); None when the graph was
    not kept."""
    try:
        raw = g.raw_cuda_graph()
    except Exception:       # noqa: BLE001 -- not kept
        return None
    rt = C.CDLL('libamdhip64.so')
    n = C.c_size_t(0)
    if rt.hipGraphGetNodes(C.c_void_p(raw), None, C.byref(n)) != 0:
        return None
    nodes = (C.c_void_p * n.value)()
    if rt.hipGraphGetNodes(C.c_void_p(raw), nodes, C.byref(n)) != 0:
        return None
    k = 0
    ty = C.c_int(0)
    for i in range(n.value):
        if rt.hipGraphNodeGetType(C.c_void_p(nodes[i]), C.byref(ty)) == 0 and ty.value == 0:     # hipGraphNodeTypeKernel
            k += 1
    return k


# Branch marks (profiling aid, scripts/branch_marks.py): when MARKS is a dict, mark(name) launches a one-lane kernel on the
# current stream that stores the device wall clock into a slot of its own -- captured into the step graphs, the slots show
# after a replay when each branch of the graph really ran.
MARKS = None
_mark_buf = None


def mark(name):
    global _mark_buf
    if MARKS is None:
        return
    if _mark_buf is None:
        _mark_buf = torch.zeros(256, dtype=torch.int64, device='cuda')
    if name not in MARKS:
        MARKS[name] = len(MARKS)
    check(lib().ssc_timestamp(C.c_void_p(_mark_buf.data_ptr() + 8 * MARKS[name]), stream_ptr()), 'ssc_timestamp')


def read_marks():
    """{name: microseconds since the earliest mark} of the last run / replay."""
    v = _mark_buf.cpu().tolist()
    t = {n: v[i] for n, i in MARKS.items() if v[i] != 0}
    t0 = min(t.values())
    return {n: (x - t0) / 100.0 for n, x in sorted(t.items(), key=lambda kv: kv[1])}


def fill(t, value):
    check(lib().ssc_fill(ptr(t), float(value), t.numel(), stream_ptr()), 'fill')


def bn_stats(x2d, scale, offset, ab, stats, eps=1e-5):
    """x2d [M, C] rows -> ab=[a;b] (y=a*x+b), stats=[mean;rstd]."""
    M, Cc = x2d.shape
    ws = workspace()
    check(lib().ssc_bn_stats(ptr(x2d), M, Cc, x2d.stride(0), ptr(scale), ptr(offset), eps, ptr(ab), ptr(stats),
                             ptr(ws), ws.numel() * 4, stream_ptr()), 'bn_stats')




class ApplyJob(object):
    """A norm backward whose sums are taken (coef written) and whose streaming apply pass is still to run (``apply_now``)."""

    def __init__(self, c, keep):
        self.c, self.keep, self.done = c, keep, False


def apply_now(job):
    """The apply pass of a deferred norm backward."""
    if job is not None and not job.done:
        job.done = True
        check(lib().ssc_bn_bwd_apply(C.byref(job.c), stream_ptr()), 'ssc_bn_bwd_apply')


def bn_act_backward(x2d, ab, stats, g1, act1, dx, g2=None, act2=ACT_NONE, dscale=None, doffset=None, pre=None, rowb=None,
                    defer=False, coef=None):
    """Backward through act(a*x+b) (has_bn when ab is given) for one or two consumers.  pre: BnBwdSums whose rows replace
    the pass that takes the two per-channel sums (only when every gradient source delivered its rows).
    rowb = (v [N, C], scale, P): g1[r] += v[r // P] * scale while it is read (a per-image gradient broadcast over P pixels).
    defer=True (needs ``coef`` [2C] when ab is given): only the sums are taken now; returns the ApplyJob of the streaming pass."""
    M, Cc = x2d.shape
    ws = workspace()
    has_bn = ab is not None
    rows, nrows = None, 0
    if pre is not None and has_bn and pre.missed == 0 and pre.sources == (1 if g2 is None else 2) and pre.rows > 0:
        rows, nrows = pre.buf, pre.rows
    if defer:
        j = BnApplyJob()
        j.x, j.M, j.C, j.ldx = x2d.data_ptr(), M, Cc, x2d.stride(0)
        j.ab, j.stats = (ab.data_ptr(), stats.data_ptr()) if has_bn else (None, None)
        j.g1, j.ldg1, j.act1 = g1.data_ptr(), g1.stride(0), act1
        j.g2, j.ldg2, j.act2 = (g2.data_ptr(), g2.stride(0), act2) if g2 is not None else (None, 0, act2)
        j.has_bn = int(has_bn)
        j.rowb, j.rowb_scale, j.rowb_P = (rowb[0].data_ptr(), float(rowb[1]), int(rowb[2])) if rowb else (None, 0.0, 0)
        j.lddx, j.dx = dx.stride(0), dx.data_ptr()
        if has_bn:
            assert coef is not None and coef.numel() == 2 * Cc and coef.is_contiguous()
            j.coef = coef.data_ptr()
            check(lib().ssc_bn_bwd_sums(C.byref(j), ptr(rows), nrows, ptr(coef), ptr(dscale), ptr(doffset), ptr(ws),
                                        ws.numel() * 4, stream_ptr()), 'ssc_bn_bwd_sums')
        return ApplyJob(j, (x2d, ab, stats, g1, g2, dx, coef, rowb))
    check(lib().ssc_bn_act_backward_pre(ptr(x2d), M, Cc, x2d.stride(0), ptr(ab), ptr(stats),
                                        ptr(g1), g1.stride(0), act1, ptr(g2), (g2.stride(0) if g2 is not None else 0),
                                        act2, int(has_bn), ptr(dx), dx.stride(0), ptr(dscale), ptr(doffset),
                                        ptr(rows), nrows, ptr(rowb[0]) if rowb else None, float(rowb[1]) if rowb else 0.0,
                                        int(rowb[2]) if rowb else 0, ptr(ws), ws.numel() * 4, stream_ptr()), 'bn_act_backward')


# ---------------------------------------------------------------------------
# thin typed wrappers over the remaining entry points
# ---------------------------------------------------------------------------
def call(name, *args):
    """Invoke a C-ABI entry point with the current stream appended; raise on error."""
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            conv.append(ptr(a))
        else:
            conv.append(a)
    check(getattr(lib(), name)(*conv, stream_ptr()), name)


def lstm_bf(C, ldk):
    """Does the recurrent step of a [C, 4C] kernel (row stride ldk) run on the bf16 pipe?"""
    return ARITH_BF16 and C % 128 == 0 and ldk == 4 * C


def lstm_hplanes_floats(rows, C):
    """Size (in fp32 elements, for the buffer pools) of the bf16 planes of an h state [rows, C] (ssc_lstm_step_fwd_bf)."""
    return ((rows + 63) // 64) * 2 * (C // 16) * 3 * 256         # whole 64-row workgroup tiles: the kernel loads both 32-row blocks


def lstm_step_fwd(h_in, Kh, ldk, g1, g2, div2, mask, mdiv, c_in, rows, C, with_gemm, c_out, h_out, acts, exact=False, hp_in=None,
                  hp_out=None):
    """One recurrent step (GEMM + gate math) in one launch; counted with the implicit-GEMM launches when profiling.
    Default arithmetic: h @ Kh as six bf16 products per fp32 product on the planes of Kh and of h (``exact`` / SSC_ARITH=fp32:
    the exact-fp32 MFMA kernels).  hp_in / hp_out: plane buffers of h_in / h_out (lstm_hplanes_floats); a loop over steps hands
    each step's hp_out to the next as hp_in, a lone call lets h_in be split here."""
    bf = lstm_bf(C, ldk) and not exact and Kh.is_contiguous()
    if bf:
        e = filter_split(Kh.view(1, 1, C, 4 * C), 0)
        if with_gemm and hp_in is None:
            hp_in = torch.empty(lstm_hplanes_floats(rows, C), device=h_in.device)
            call('ssc_lstm_hsplit', h_in, rows, C, hp_in)
        name = 'ssc_lstm_step_fwd_bf'
        args = (h_in, hp_in if with_gemm else None, e.buf, e.nbp, g1, g2, div2, mask, mdiv, c_in, rows, C, c_out, h_out, hp_out, acts)
    else:
        name, args = 'ssc_lstm_step_fwd', (h_in, Kh, ldk, g1, g2, div2, mask, mdiv, c_in, rows, C, int(with_gemm), c_out, h_out, acts)
    if PROFILE is None or not with_gemm:
        call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(name, *args)
    e1.record()
    PROFILE.append(('lstm_step_fwd_bf16x6<64x32>' if bf else 'lstm_step_fwd<64x64>', 2.0 * rows * C * 4 * C, e0, e1, (rows, 4 * C, C),
                    (6.0 if bf else 4.0) * (rows * C + C * 4 * C) + 4.0 * 3 * rows * 4 * C))


def concat_parts(out, parts):
    """out [N,H,W,ldo] <- channel concat of up to three parts; part = dict(x=tensor [N,h,w,ld], C=real channels,
    ab=None | [2C] | [N,2C], act=ACT_*, upsample=bool, gate=None | (raw gate [N,H,W,C], mnmx [N,2,C]))."""
    N, H, W, ldo = out.shape
    d = CatDesc()
    d.out, d.nparts, d.ldo, d.N, d.H, d.W = out.data_ptr(), len(parts), ldo, N, H, W
    tot = 0
    for k, q in enumerate(parts):
        x = q['x']
        cp = d.p[k]
        cp.x, cp.ld, cp.C = x.data_ptr(), x.shape[-1], q.get('C', x.shape[-1])
        ab = q.get('ab')
        cp.ab = ab.data_ptr() if ab is not None else None
        cp.ab_sample_stride = 2 * cp.C if (ab is not None and ab.dim() == 2) else 0
        cp.act, cp.upsample = q.get('act', ACT_NONE), int(bool(q.get('upsample', False)))
        assert x.shape[1] * (2 if cp.upsample else 1) == H
        g = q.get('gate')
        cp.gate, cp.mnmx = (g[0].data_ptr(), g[1].data_ptr()) if g is not None else (None, None)
        assert g is None or g[0].shape[-1] == cp.C
        tot += cp.C
    assert tot <= ldo
    check(lib().ssc_concat_parts(C.byref(d), stream_ptr()), 'ssc_concat_parts')


def minmax_hw(x4d, mnmx):
    """mnmx [N,2,C] <- per-sample per-channel min / max over H*W of x4d [N,H,W,C]."""
    n, h, w, c = x4d.shape
    ws = workspace()
    check(lib().ssc_minmax_hw(ptr(x4d), c, n, h * w, c, ptr(mnmx), ptr(ws), ws.numel() * 4, stream_ptr()), 'minmax_hw')


def bn_stats_view(x4d, scale, offset, ab, stats, eps=1e-5):
    n, h, w, c = x4d.shape
    bn_stats(x4d.view(n * h * w, c), scale, offset, ab, stats, eps)
