"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/pmoe_hip.h).

Activations are NHWC tensors ``[N, H, W, C]`` (bf16 or f32) with the experts folded into N
(image n belongs to expert ``n // ipe``).  These functions only validate and launch; they never
compute anything with torch ops.
"""
import ctypes as C

import torch

from . import hip
from .hip import ConvDesc, WgradDesc, check, dt, load, ptr, stream_ptr


def _round_up(v, m):
    return (v + m - 1) // m * m


def conv_out_size(h, ks, stride, pad):
    return (h + 2 * pad - ks) // stride + 1


def _nhwc(t, name):
    if t.dim() != 4:
        raise ValueError(f"{name}: expected NHWC [N,H,W,C], got shape {tuple(t.shape)}")
    return t.shape


def conv2d(x, w_packed, out, *, cin, cout, coutp, ipe, ks, stride, pad, dilate=False, in_shared=False,
           in_coff=0, out_coff=0, res=None, res_coff=0, res_mode=hip.RES_NONE, bias=None, act=hip.ACT_NONE,
           drop_p=0.0, seed=0, stats=None, plan_only=False, out_scale=None, in_scale=1.0, bn_coef=None, bn_ipe=0, shuffle2_c=0):
    """out[..., out_coff:out_coff+cout] = epilogue(conv(x[..., in_coff:in_coff+cin], w)).
    ``x`` [Nin,H,W,ldx], ``out`` [N,Ho,Wo,ldo] preallocated; also used for dgrad and grouped GEMM.
    ``plan_only``: launch nothing, return the kernel-instantiation code of ``pmoe_conv2d_plan`` (include/pmoe_hip.h)."""
    nin, h, w_, ldx = _nhwc(x, "x")
    n, ho, wo, ldo = _nhwc(out, "out")
    if shuffle2_c:
        # ConvTranspose2d(k2, s2) in one launch (pmoe_conv_desc.shuffle_c): ``out`` is the DESTINATION [N, 2 Ho, 2 Wo, ld]; the 1x1
        # convolution's own output map has half its sides
        if ho % 2 or wo % 2 or ks != 1 or cout != 4 * shuffle2_c:
            raise ValueError("conv2d: shuffle2_c needs a 1x1 layer with 4 * shuffle2_c outputs and an even-sided destination")
        ho, wo = ho // 2, wo // 2
    w_fp8 = w_packed.dtype == torch.uint8          # e4m3 bytes (pack_conv_weights_fp8): BASELINE config 5
    in_fp8 = w_fp8 and x.dtype == torch.uint8      # ... and e4m3 activations (bn_apply's fp8 side output): the block-scaled MFMA kernel
    if (x.dtype != out.dtype and not in_fp8) or (w_packed.dtype != out.dtype and not w_fp8):
        raise ValueError("conv2d: x, w and out must share one dtype")
    if w_fp8 and (out_scale is None or out.dtype != torch.bfloat16):
        raise ValueError("conv2d: e4m3 weights need bf16 activations and the per-channel out_scale of the pack")
    d = ConvDesc()
    d.in_, d.w, d.out = ptr(x, "x"), ptr(w_packed, "w"), ptr(out, "out")
    d.res = ptr(res, "res", out.dtype) if res is not None else None
    d.bias = ptr(bias, "bias", torch.float32) if bias is not None else None
    d.stats = ptr(stats, "stats", torch.float32) if stats is not None else None
    d.n, d.h, d.w_, d.cin = n, h, w_, cin
    d.ho, d.wo, d.cout, d.coutp = ho, wo, cout, coutp
    d.in_ld, d.in_coff, d.out_ld, d.out_coff = ldx, in_coff, ldo, out_coff
    d.res_ld = res.shape[-1] if res is not None else 0
    d.res_coff = res_coff
    d.ipe, d.in_shared = ipe, int(in_shared)
    d.ks, d.stride, d.pad, d.dilate = ks, stride, pad, int(dilate)
    d.act, d.res_mode = act, res_mode if (res is not None or res_mode == hip.RES_INBN) else hip.RES_NONE
    d.drop_p, d.seed, d.dtype = float(drop_p), int(seed), dt(out)
    d.shuffle_c = int(shuffle2_c)
    if w_fp8:
        d.w_fp8, d.in_scale, d.out_scale = 1, float(in_scale), ptr(out_scale, "out_scale", torch.float32)
        d.in_fp8 = int(in_fp8)
    if d.res_mode == hip.RES_DBN:
        # data gradient into relu(BatchNorm(z)): res = z, bn_coef = [4][n / bn_ipe][cout] (mean, invstd, gamma*invstd, beta);
        # the launch masks the gradient and leaves the BatchNorm backward's channel reductions in `stats`
        nset = n // (bn_ipe or ipe)
        if bn_coef is None or bn_coef.dtype != torch.float32 or tuple(bn_coef.shape) != (4, nset, cout):
            raise ValueError(f"conv2d: RES_DBN needs bn_coef [4, {nset}, {cout}] f32")
        if stats is None and not plan_only:
            raise ValueError("conv2d: RES_DBN writes the BatchNorm-backward reductions to `stats`")
        d.bn_coef, d.bn_ipe = ptr(bn_coef, "bn_coef", torch.float32), int(bn_ipe or ipe)
    if d.res_mode == hip.RES_INBN:
        # the INPUT is the pre-activation z of a BatchNorm + ReLU: bn_coef = [4][n / bn_ipe][cin], applied on load
        nset = n // (bn_ipe or ipe)
        if res is not None or bn_coef is None or bn_coef.dtype != torch.float32 or tuple(bn_coef.shape) != (4, nset, cin):
            raise ValueError(f"conv2d: RES_INBN takes no `res` and needs bn_coef [4, {nset}, {cin}] f32")
        d.bn_coef, d.bn_ipe = ptr(bn_coef, "bn_coef", torch.float32), int(bn_ipe or ipe)
    if in_shared and nin != ipe:
        raise ValueError("conv2d: shared input must hold exactly ipe images")
    if not in_shared and nin != n:
        raise ValueError("conv2d: input/output image counts differ")
    if w_packed.numel() < (n // ipe) * coutp * ks * ks * cin:
        raise ValueError("conv2d: packed weight tensor too small")
    if plan_only:
        return load().pmoe_conv2d_plan(C.byref(d))
    if stats is not None:
        # the launch writes exactly pmoe_conv2d_stat_rows(d) partial-sum rows: a buffer sized from a different descriptor
        # (other row lengths can mean another kernel) would be folded with unwritten rows
        rows = load().pmoe_conv2d_stat_rows(C.byref(d))
        if stats.dim() != 3 or stats.shape[0] != rows or stats.shape[1] != 2 or stats.shape[2] != coutp:
            raise ValueError(f"conv2d: stats must be [{rows}, 2, {coutp}] for this launch, got {tuple(stats.shape)}")
    if _prof is not None:            # profiling: remember which kernel instantiation serves this launch
        _launch_info["kernel"] = load().pmoe_conv2d_plan(C.byref(d))
    check(load().pmoe_conv2d_igemm(C.byref(d), stream_ptr()), "pmoe_conv2d_igemm")
    return out


def conv2d_stat_rows(n, h, w_, ho, wo, cin, cout, coutp, ipe, ks, stride, pad, dtype, w_fp8=False, in_ld=0, out_ld=0,
                     in_shared=False, in_fp8=False):
    """Partial-sum rows the launch will write.  in_ld / out_ld: row lengths (elements) of the tensors the launch will get --
    the kernel choice can depend on them (0 = dense)."""
    d = ConvDesc()
    d.w_fp8, d.in_scale, d.in_fp8 = int(w_fp8), 1.0, int(in_fp8)
    if in_fp8:
        d.out_scale = C.c_void_p(1)          # (planning only: the fp8 kernel's plan asks for a scale pointer, nothing dereferences it)
    d.in_ld, d.out_ld, d.in_shared = in_ld, out_ld, int(in_shared)
    d.n, d.h, d.w_, d.cin, d.ho, d.wo, d.cout, d.coutp = n, h, w_, cin, ho, wo, cout, coutp
    d.ipe, d.ks, d.stride, d.pad, d.dtype = ipe, ks, stride, pad, hip._TORCH_DT[dtype]
    rows = load().pmoe_conv2d_stat_rows(C.byref(d))
    if rows < 0:
        check(rows, "pmoe_conv2d_stat_rows")
    return rows


_wgrad_scratch = {}


def _wgrad_part_ws(device, floats):
    """K-split scratch of the weight-gradient launch: one growing f32 buffer per (device, stream) -- launches on one
    stream are ordered, so consecutive layers may share it."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _wgrad_scratch.get(key)
    if buf is None or buf.numel() < floats:
        buf = _wgrad_scratch[key] = torch.empty(max(floats, 1 << 22), dtype=torch.float32, device=device)
    return buf


def conv2d_wgrad(x, dy, dw_ws, *, cin, cout, cinp, coutp, ipe, ks, stride, pad, x_shared=False, x_coff=0, dy_coff=0,
                 per_image=False, plan_only=False, grads=None, grads_cout=0, grads_cin=0, defer_fold=False, bn_fuse=None):
    """dw_ws [E | N][ks*ks][coutp][cinp] f32 is OVERWRITTEN with the weight gradient (deterministic: fixed-order folds
    of the pixel split, no atomics; csrc/conv_wgrad.hip).  ``grads`` (flat f32, E * grads_cout * grads_cin * ks * ks): the
    parameters' own gradient layout, written by the fold instead of dw_ws (which then is scratch only).  ``defer_fold``: run the
    MFMA launch only and return the descriptor for :func:`conv2d_wgrad_fold` (the two kernels are then timed apart)."""
    nin, h, w_, ldx = _nhwc(x, "x")
    n, ho, wo, ldy = _nhwc(dy, "dy")
    d = WgradDesc()
    d.x, d.dy = ptr(x, "x"), ptr(dy, "dy", x.dtype)
    d.dw_ws = ptr(dw_ws, "dw_ws", torch.float32) if dw_ws is not None else None
    if bn_fuse is not None:
        # ``dy`` is g, the masked gradient w.r.t. the output of the BatchNorm + ReLU behind this conv; bn_fuse = (z, coef
        # [4,E,C], c1 [E,C], c2 [E,C]): the kernel applies that BatchNorm's backward on load (include/pmoe_hip.h, bn_fused)
        z, coef, c1, c2 = bn_fuse
        if z.shape[:3] != dy.shape[:3]:
            raise ValueError("conv2d_wgrad: bn_fuse z must have the geometry of dy")
        d.bn_fused, d.bn_z_ld = 1, z.shape[-1]
        if not plan_only:
            E_ = n // ipe
            if coef.shape != (4, E_, cout) or c1.shape != (E_, cout) or c2.shape != (E_, cout):
                raise ValueError("conv2d_wgrad: bn_fuse coefficient blocks must be [4,E,cout] / [E,cout]")
            d.bn_z, d.bn_coef = ptr(z, "bn_z", x.dtype), ptr(coef, "bn_coef", torch.float32)
            d.bn_c1, d.bn_c2 = ptr(c1, "bn_c1", torch.float32), ptr(c2, "bn_c2", torch.float32)
    d.n, d.h, d.w_, d.cin, d.cinp = n, h, w_, cin, cinp
    d.ho, d.wo, d.cout, d.coutp = ho, wo, cout, coutp
    d.x_ld, d.x_coff, d.dy_ld, d.dy_coff = ldx, x_coff, ldy, dy_coff
    d.ipe, d.x_shared = ipe, int(x_shared)
    d.ks, d.stride, d.pad, d.dtype = ks, stride, pad, dt(x)
    d.per_image = int(per_image)
    if grads is not None:
        if per_image or grads.dtype != torch.float32 or grads.numel() != (n // ipe) * grads_cout * grads_cin * ks * ks:
            raise ValueError("conv2d_wgrad: grads must hold E * cout * cin * ks * ks float32 values (not with per_image)")
        d.grads, d.cout_real, d.cin_real = ptr(grads, "grads", torch.float32), int(grads_cout), int(grads_cin)
    if bn_fuse is not None and plan_only:       # -> 7209 if the fused kernel serves this shape, else a negative error code
        return load().pmoe_conv2d_wgrad_plan(C.byref(d))
    if dw_ws.numel() < (n if per_image else n // ipe) * ks * ks * coutp * cinp:
        raise ValueError("conv2d_wgrad: workspace too small")
    need = load().pmoe_conv2d_wgrad_ws_floats(C.byref(d))
    if need < 0:
        check(int(need), "pmoe_conv2d_wgrad_ws_floats")
    if plan_only:                    # number of K-split slices (workgroups along the pixel axis) the launch would use
        return max(1, need // ((n // ipe) * ks * ks * coutp * cinp))
    if need > 0:
        part = _wgrad_part_ws(x.device, need)
        d.part_ws, d.part_ws_floats = ptr(part, "part_ws"), part.numel()
    if _prof is not None:            # profiling: remember which kernel instantiation serves this launch
        _launch_info["kernel"] = load().pmoe_conv2d_wgrad_plan(C.byref(d))
    d.defer_fold = int(defer_fold)
    check(load().pmoe_conv2d_wgrad(C.byref(d), stream_ptr()), "pmoe_conv2d_wgrad")
    return d if defer_fold else dw_ws


def mlp_wgrad(x, dy, grads, bias_grads, *, cin, cout, cin_real, cout_real, ipe, x_shared=False, x_coff=0, dy_coff=0):
    """weight (+ bias) gradient of one Linear layer of all experts, in the parameters' own [E][out][in] / [E][out] layout
    (csrc/gemm_skinny.hip: mlp_wgrad_kernel).  x [Nx,1,1,ldx] / dy [N,1,1,ldy]: bf16 rows."""
    n, ldy = dy.shape[0], dy.shape[-1]
    E = n // ipe
    if x.dtype != torch.bfloat16 or dy.numel() != n * ldy or x.numel() != x.shape[0] * x.shape[-1]:
        raise ValueError("mlp_wgrad: bf16 rows [N,1,1,ld]")
    if grads.numel() != E * cout_real * cin_real or (bias_grads is not None and bias_grads.numel() != E * cout_real):
        raise ValueError("mlp_wgrad: grads must hold E * out * in (bias_grads: E * out) float32 values")
    check(load().pmoe_mlp_wgrad(ptr(x, "x"), ptr(dy, "dy", x.dtype), ptr(grads, "grads", torch.float32),
                                ptr(bias_grads, "bias_grads", torch.float32), n, ipe, int(x_shared), cin, cout, cin_real,
                                cout_real, x.shape[-1], x_coff, ldy, dy_coff, dt(x), stream_ptr()), "pmoe_mlp_wgrad")


def conv2d_wgrad_fold(d):
    """the deferred tail of :func:`conv2d_wgrad` (``defer_fold``): K-split slabs -> the parameter-layout gradient / dw_ws"""
    check(load().pmoe_conv2d_wgrad_fold(C.byref(d), stream_ptr()), "pmoe_conv2d_wgrad_fold")


def pack_conv_weights(ptr_tab, fwd, dgrd, E, cout, cin, ks, coutp, cinp, cinp2, coutp2, dtype):
    check(load().pmoe_pack_conv_weights(ptr(ptr_tab, "ptr table", torch.int64), ptr(fwd), ptr(dgrd), E, cout, cin, ks,
                                        coutp, cinp, cinp2, coutp2, hip._TORCH_DT[dtype], stream_ptr()),
          "pmoe_pack_conv_weights")


def pack_conv_weights_fp8(ptr_tab, fwd, dgrd, wscale, oscale, in_scale, E, cout, cin, ks, coutp, cinp, cinp2, coutp2):
    """e4m3 weight pack with one power-of-two scale per output channel (include/pmoe_hip.h)."""
    check(load().pmoe_pack_conv_weights_fp8(ptr(ptr_tab, "ptr table", torch.int64), ptr(fwd, "fwd", torch.uint8),
                                            ptr(dgrd, "dgrd", torch.bfloat16) if dgrd is not None else None,
                                            ptr(wscale, "wscale", torch.float32), ptr(oscale, "oscale", torch.float32),
                                            float(in_scale), E, cout, cin, ks, coutp, cinp, cinp2, coutp2, stream_ptr()),
          "pmoe_pack_conv_weights_fp8")


def pack_conv_weights_scaled(ptr_tab, scale, shift, mean, fwd, bias, E, cout, cin, ks, coutp, cinp, dtype):
    f32 = torch.float32
    check(load().pmoe_pack_conv_weights_scaled(ptr(ptr_tab, "ptr table", torch.int64), ptr(scale, "scale", f32),
                                               ptr(shift, "shift", f32), ptr(mean, "mean", f32), ptr(fwd),
                                               ptr(bias, "bias", f32), E, cout, cin, ks, coutp, cinp, hip._TORCH_DT[dtype],
                                               stream_ptr()), "pmoe_pack_conv_weights_scaled")


def pack_conv_weights_gated(ptr_tab, gate, fwd, dgrd, N, ipe, cout, cin, ks, coutp, cinp, cinp2, coutp2, dtype):
    check(load().pmoe_pack_conv_weights_gated(ptr(ptr_tab, "ptr table", torch.int64), ptr(gate, "gate", torch.float32),
                                              gate.shape[-1], ptr(fwd), ptr(dgrd), N, ipe, cout, cin, ks, coutp, cinp, cinp2,
                                              coutp2, hip._TORCH_DT[dtype], stream_ptr()), "pmoe_pack_conv_weights_gated")


def unpack_conv_wgrad(dw_ws, grads, E, cout, cin, ks, coutp, cinp):
    check(load().pmoe_unpack_conv_wgrad(ptr(dw_ws, "dw_ws", torch.float32), ptr(grads, "grads", torch.float32), E, cout,
                                        cin, ks, coutp, cinp, stream_ptr()), "pmoe_unpack_conv_wgrad")


def pack_bias(ptr_tab, dst, E, cout, coutp):
    check(load().pmoe_pack_bias(ptr(ptr_tab, "ptr table", torch.int64), ptr(dst, "dst", torch.float32), E, cout, coutp,
                                stream_ptr()), "pmoe_pack_bias")


def act_fwd(x, y, act, drop_p=0.0, seed=0):
    check(load().pmoe_act_fwd(ptr(x, "x"), ptr(y, "y", x.dtype), x.numel(), act, float(drop_p), int(seed), dt(x), stream_ptr()),
          "pmoe_act_fwd")


def act_bwd(dy, y, dx, act, drop_p=0.0):
    check(load().pmoe_act_bwd(ptr(dy, "dy"), ptr(y, "y", dy.dtype), ptr(dx, "dx", dy.dtype), dy.numel(), act, float(drop_p),
                              dt(dy), stream_ptr()), "pmoe_act_bwd")


def colstats(x2d_rows_per_expert, x, E, C_, part, nparts, ld=None, coff=0, shiftc=None):
    """partial sums of (x - c), (x - c)^2 with c = row 0 of each expert (stored to shiftc [E,C]); c = 0 if None."""
    check(load().pmoe_colstats(ptr(x, "x"), x2d_rows_per_expert, E, C_, ld if ld is not None else x.shape[-1], coff,
                               ptr(part, "part", torch.float32), nparts, ptr(shiftc, "shiftc", torch.float32), dt(x),
                               stream_ptr()), "pmoe_colstats")


def reduce_partials(part_in, part_out, E, nin, nout, width):
    check(load().pmoe_reduce_partials(ptr(part_in, "part_in", torch.float32), ptr(part_out, "part_out", torch.float32),
                                      E, nin, nout, width, stream_ptr()), "pmoe_reduce_partials")


def bn_finalize(part, nparts, count, gamma_tab, beta_tab, rmean_tab, rvar_tab, momentum, eps, training, scale, shift,
                mean, invstd, E, C_, shiftc=None):
    f32 = torch.float32
    check(load().pmoe_bn_finalize(ptr(part, "part", f32), nparts, count, ptr(gamma_tab), ptr(beta_tab), ptr(rmean_tab),
                                  ptr(rvar_tab), momentum, eps, int(training), ptr(scale, "scale", f32),
                                  ptr(shift, "shift", f32), ptr(mean, "mean", f32), ptr(invstd, "invstd", f32), E, C_,
                                  ptr(shiftc, "shiftc", f32), stream_ptr()), "pmoe_bn_finalize")


def bn_apply(x, res, y, scale, shift, mean, rpe, E, C_, relu, y_coff=0, y_fp8=None, in_scale=1.0):
    """y[..., y_coff:y_coff+C] = [relu]((x - mean)*scale + shift [+ res]); shift is the BN beta (bn_finalize's output).
    ``y_fp8`` (uint8, same shape as a dense y): also e4m3(bf16(y) * in_scale), the operand of the block-scaled fp8 conv."""
    if y_fp8 is not None and (y_fp8.dtype != torch.uint8 or y_fp8.shape != y.shape):
        raise ValueError("bn_apply: y_fp8 must be a uint8 tensor of y's shape")
    check(load().pmoe_bn_apply(ptr(x, "x"), ptr(res, "res", x.dtype), ptr(y, "y", x.dtype), ptr(scale), ptr(shift),
                               ptr(mean, "mean", torch.float32), rpe,
                               E, C_, int(relu), y.shape[-1], y_coff, dt(x), ptr(y_fp8, "y_fp8", torch.uint8),
                               float(in_scale), stream_ptr()), "pmoe_bn_apply")


def bn_apply_pool2(x, y, pooled, scale, shift, mean, ipe, E, C_, relu, y_coff=0):
    """y[..., y_coff:y_coff+C] = [relu]((x - mean)*scale + shift) and pooled = MaxPool2d(2,2)(that), one pass (U-Net down blocks)."""
    n, h, w_, c = _nhwc(x, "x")
    if c != C_ or tuple(pooled.shape) != (n, h // 2, w_ // 2, C_) or tuple(y.shape[:3]) != (n, h, w_) or n != E * ipe:
        raise ValueError("bn_apply_pool2: x [E*ipe,H,W,C], y [E*ipe,H,W,ld], pooled [E*ipe,H/2,W/2,C]")
    check(load().pmoe_bn_apply_pool2(ptr(x, "x"), ptr(y, "y", x.dtype), ptr(pooled, "pooled", x.dtype), ptr(scale), ptr(shift),
                                     ptr(mean, "mean", torch.float32), ipe, h, w_, E, C_, int(relu), y.shape[-1], y_coff, dt(x),
                                     stream_ptr()), "pmoe_bn_apply_pool2")


def bn_bwd_reduce(dy, y, x, mean, invstd, scale, shift, rpe, E, C_, relu, part, nparts, gmask=None):
    """y=None (relu, no residual in forward): the ReLU mask is recomputed from x, the saved output is not read.
    gmask: the masked gradient is stored there (bn_bwd_apply then takes it as dy with relu=False, y=None)."""
    check(load().pmoe_bn_bwd_reduce(ptr(dy, "dy"), ptr(y, "y", dy.dtype), ptr(x, "x", dy.dtype), ptr(mean), ptr(invstd),
                                    ptr(scale), ptr(shift), rpe, E, C_, int(relu), ptr(part, "part", torch.float32), nparts,
                                    ptr(gmask, "gmask", dy.dtype), dt(dy), stream_ptr()), "pmoe_bn_bwd_reduce")


def bn_bwd_finalize(part, nparts, count, dgamma, dbeta, c1, c2, E, C_):
    check(load().pmoe_bn_bwd_finalize(ptr(part), nparts, count, ptr(dgamma), ptr(dbeta), ptr(c1), ptr(c2), E, C_,
                                      stream_ptr()), "pmoe_bn_bwd_finalize")


def bn_bwd_apply(dy, y, x, mean, invstd, scale, shift, c1, c2, dx, gmask, rpe, E, C_, relu):
    check(load().pmoe_bn_bwd_apply(ptr(dy, "dy"), ptr(y, "y", dy.dtype), ptr(x, "x", dy.dtype), ptr(mean), ptr(invstd),
                                   ptr(scale), ptr(shift), ptr(c1), ptr(c2), ptr(dx, "dx", dy.dtype), ptr(gmask, "gmask", dy.dtype),
                                   rpe, E, C_, int(relu), dt(dy), stream_ptr()), "pmoe_bn_bwd_apply")


def stem_tail_stats(z2, sc2, sh2, mu2, part, nparts, E, ipe, shiftc=None, part_x=None):
    n, h, w_, c = _nhwc(z2, "z2")
    check(load().pmoe_stem_tail_stats(ptr(z2, "z2"), ptr(sc2), ptr(sh2), ptr(mu2), ptr(part, "part", torch.float32), nparts,
                                      ptr(shiftc, "shiftc", torch.float32), ptr(part_x, "part_x", torch.float32), E, ipe,
                                      h, w_, c, dt(z2), stream_ptr()), "pmoe_stem_tail_stats")


def _consts12(consts):
    return (C.c_void_p * 12)(*[t.data_ptr() if t is not None else None for t in consts])


def stem_tail_pooled(y, dpool, argmax, consts, part4, nparts, E):
    """train-mode backward over the pooled tensors: part4 [E,nparts,4,C] = sums of g, g*xhat1, g*m, g*m*xhat2."""
    n, ho, wo, c = _nhwc(y, "y")
    check(load().pmoe_stem_tail_pooled(ptr(y, "y"), ptr(dpool, "dpool", y.dtype), ptr(argmax, "argmax", torch.uint8),
                                       _consts12(consts), ptr(part4, "part4", torch.float32), nparts, E,
                                       (n // E) * ho * wo, c, dt(y), stream_ptr()), "pmoe_stem_tail_pooled")


def stem_tail_combine(part4, np4, part_x, npx, consts, count, out1, out2, E, c):
    f32 = torch.float32
    check(load().pmoe_stem_tail_combine(ptr(part4, "part4", f32), np4, ptr(part_x, "part_x", f32), npx, _consts12(consts),
                                        count, ptr(out1, "out1", f32), ptr(out2, "out2", f32), E, c, stream_ptr()),
          "pmoe_stem_tail_combine")


def stem_tail_pool(z2, y, argmax, sc2, sh2, sc1, sh1, mu2, mu1, ipe):
    n, h, w_, c = _nhwc(z2, "z2")
    check(load().pmoe_stem_tail_pool(ptr(z2, "z2"), ptr(y, "y", z2.dtype), ptr(argmax, "argmax", torch.uint8), ptr(sc2),
                                     ptr(sh2), ptr(sc1), ptr(sh1), ptr(mu2), ptr(mu1), n, ipe, h, w_, c, dt(z2), stream_ptr()),
          "pmoe_stem_tail_pool")


def stem_tail_bwd(phase, z2, dpool, argmax, dz2, consts, part, nparts, E, ipe):
    """consts: list of 12 f32 [E,C] tensors (or None): sc2 sh2 sc1 sh1 mu1 is1 mu2 is2 c11 c21 c12 c22."""
    n, h, w_, c = _nhwc(z2, "z2")
    arr = (C.c_void_p * 12)(*[t.data_ptr() if t is not None else None for t in consts])
    check(load().pmoe_stem_tail_bwd(phase, ptr(z2, "z2"), ptr(dpool, "dpool", z2.dtype),
                                    ptr(argmax, "argmax", torch.uint8), ptr(dz2, "dz2", z2.dtype), arr,
                                    ptr(part, "part", torch.float32), nparts, E, ipe, h, w_, c, dt(z2), stream_ptr()),
          "pmoe_stem_tail_bwd")


def maxpool_fwd(x, y, argmax):
    n, h, w_, c = _nhwc(x, "x")
    check(load().pmoe_maxpool3s2_fwd(ptr(x, "x"), ptr(y, "y", x.dtype), ptr(argmax, "argmax", torch.uint8), n, h, w_, c,
                                     dt(x), stream_ptr()), "pmoe_maxpool3s2_fwd")


def maxpool_bwd(dy, argmax, dx):
    n, h, w_, c = _nhwc(dx, "dx")
    check(load().pmoe_maxpool3s2_bwd(ptr(dy, "dy"), ptr(argmax, "argmax", torch.uint8), ptr(dx, "dx", dy.dtype), n, h,
                                     w_, c, dt(dy), stream_ptr()), "pmoe_maxpool3s2_bwd")


def gap_partial(a, b, part, nparts, b_shared_ipe=0):
    n, h, w_, c = _nhwc(a, "a")
    check(load().pmoe_gap_partial(ptr(a, "a"), ptr(b, "b", a.dtype), ptr(part, "part", torch.float32), n, h * w_, c,
                                  nparts, b_shared_ipe, dt(a), stream_ptr()), "pmoe_gap_partial")


def bn_apply_gap(x, y, scale, shift, mean, part, nparts, ipe, relu):
    """bn_apply (no residual) + gap_partial(y) in one pass over the activation; part [N, nparts, C] f32."""
    n, h, w_, c = _nhwc(x, "x")
    check(load().pmoe_bn_apply_gap(ptr(x, "x"), ptr(y, "y", x.dtype), ptr(scale), ptr(shift), ptr(mean, "mean", torch.float32),
                                   ptr(part, "part", torch.float32), nparts, n, ipe, h * w_, c, int(relu), dt(x), stream_ptr()),
          "pmoe_bn_apply_gap")


def gap_finish(part, out, n, c, nparts, hw, out_ld, out_coff):
    check(load().pmoe_gap_finish(ptr(part, "part", torch.float32), ptr(out, "out"), n, c, nparts, hw, out_ld, out_coff,
                                 dt(out), stream_ptr()), "pmoe_gap_finish")


def gap_bwd(g, dx, g_ld, g_coff):
    n, h, w_, c = _nhwc(dx, "dx")
    check(load().pmoe_gap_bwd(ptr(g, "g", dx.dtype), ptr(dx, "dx"), n, h * w_, c, g_ld, g_coff, dt(dx), stream_ptr()),
          "pmoe_gap_bwd")


def eca_gate(gap_part, nparts, hw, w_tab, k, gate, gapmean, n, ipe, in_ipe, c, creal):
    f32 = torch.float32
    check(load().pmoe_eca_gate(ptr(gap_part, "gap_part", f32), nparts, hw, ptr(w_tab), k, ptr(gate, "gate", f32),
                               ptr(gapmean, "gapmean", f32), n, ipe, in_ipe, c, creal, stream_ptr()), "pmoe_eca_gate")


def eca_scale(x, gate, y, x_shared_ipe=0):
    n, h, w_, c = _nhwc(y, "y")
    check(load().pmoe_eca_scale(ptr(x, "x"), ptr(gate, "gate", torch.float32), ptr(y, "y", x.dtype), n, h * w_, c,
                                x_shared_ipe, dt(x), stream_ptr()), "pmoe_eca_scale")


def eca_bwd_small(dot_part, nparts, gate, gapmean, w_tab, k, dgap, dw, n, ipe, c, creal, dgap_scale=1.0):
    scratch = torch.empty(n, k, dtype=torch.float32, device=gate.device)
    check(load().pmoe_eca_bwd_small(ptr(dot_part), nparts, ptr(gate), ptr(gapmean), ptr(w_tab), k, ptr(dgap), ptr(dw),
                                    ptr(scratch), n, ipe, c, creal, float(dgap_scale), stream_ptr()), "pmoe_eca_bwd_small")


def eca_stem_fold(G, gate, w_tab, dw, ds, n, ipe, cout, cin, ks, coutp, cinp):
    f32 = torch.float32
    check(load().pmoe_eca_stem_fold(ptr(G, "G", f32), ptr(gate, "gate", f32), ptr(w_tab), ptr(dw, "dw", f32),
                                    ptr(ds, "ds", f32), n, ipe, cout, cin, ks, coutp, cinp, gate.shape[-1],
                                    stream_ptr()),
          "pmoe_eca_stem_fold")


def eca_bwd_apply(dy, gate, dgap, dx):
    n, h, w_, c = _nhwc(dy, "dy")
    check(load().pmoe_eca_bwd_apply(ptr(dy, "dy"), ptr(gate), ptr(dgap), ptr(dx, "dx", dy.dtype), n, h * w_, c, dt(dy),
                                    stream_ptr()), "pmoe_eca_bwd_apply")


def nchw_to_nhwc(src, dst):
    b, c, h, w_ = src.shape
    check(load().pmoe_nchw_to_nhwc(ptr(src, "src", torch.float32), ptr(dst, "dst"), b, c, h, w_, dst.shape[-1], dt(dst),
                                   stream_ptr()), "pmoe_nchw_to_nhwc")


def pad_rows(src, dst):
    b, k = src.shape
    check(load().pmoe_pad_rows(ptr(src, "src", torch.float32), ptr(dst, "dst"), b, k, dst.shape[-1], dt(dst),
                               stream_ptr()), "pmoe_pad_rows")


def gate_mixture_fwd(head, spd, probs, mean, std, speeds, B, E, alpha_relu, shared=False):
    f32 = torch.float32
    check(load().pmoe_gate_mixture_fwd(ptr(head, "head"), head.shape[-1], ptr(spd, "spd", head.dtype), spd.shape[-1],
                                       ptr(probs, "probs", f32), ptr(mean, "mean", f32), ptr(std, "std", f32),
                                       ptr(speeds, "speeds", f32), B, E, int(alpha_relu), int(shared), dt(head),
                                       stream_ptr()), "pmoe_gate_mixture_fwd")


def gate_mixture_bwd(head, probs, dprobs, dmean, dstd, dspeeds, dhead, dspd, B, E, alpha_relu, shared=False):
    f32 = torch.float32
    check(load().pmoe_gate_mixture_bwd(ptr(head, "head"), head.shape[-1], ptr(probs, "probs", f32),
                                       ptr(dprobs, "dprobs", f32), ptr(dmean, "dmean", f32), ptr(dstd, "dstd", f32),
                                       ptr(dspeeds, "dspeeds", f32), ptr(dhead, "dhead", head.dtype),
                                       ptr(dspd, "dspd", head.dtype), dspd.shape[-1], B, E, int(alpha_relu), int(shared),
                                       dt(head), stream_ptr()), "pmoe_gate_mixture_bwd")


def moe_loss(probs, mean, std, speeds, actions, target, c0, c1, loss, loglik, dprobs, dmean, dstd, dspeeds, B, E,
             shared_speed=False):
    f32 = torch.float32
    check(load().pmoe_moe_loss(ptr(probs, "probs", f32), ptr(mean, "mean", f32), ptr(std, "std", f32),
                               ptr(speeds, "speeds", f32), ptr(actions, "actions", f32), ptr(target, "target", f32),
                               float(c0), float(c1), ptr(loss, "loss", f32), ptr(loglik, "loglik", f32),
                               ptr(dprobs, "dprobs", f32), ptr(dmean, "dmean", f32), ptr(dstd, "dstd", f32),
                               ptr(dspeeds, "dspeeds", f32), B, E, int(shared_speed), stream_ptr()), "pmoe_moe_loss")


# ---------------------------------------------------------------------------------------------------
# Optional per-launch timing with HIP events on the launch stream (used by bench.py for the roofline
# object and the per-kernel breakdown; off by default -> zero overhead besides one `is None` test).
import functools as _functools

_prof = None
_next_meta = {}
_launch_info = {}        # filled by a wrapper during its launch (e.g. which conv kernel instantiation ran)


def set_meta(**kw):
    """Algorithmic work (flop=..., bytes=...) of the NEXT launch, recorded only while profiling."""
    global _next_meta
    if _prof is not None:
        _next_meta = kw


def profile_begin():
    global _prof
    _prof = []


def profile_end():
    """-> [(op name, meta dict, milliseconds)] for every launch since profile_begin()."""
    global _prof
    recs, _prof = _prof, None
    torch.cuda.synchronize()
    return [(n, m, e0.elapsed_time(e1)) for n, m, e0, e1 in recs]


def _timed(fn):
    @_functools.wraps(fn)
    def wrapper(*a, **k):
        global _next_meta
        if _prof is None:
            return fn(*a, **k)
        meta, _next_meta = _next_meta, {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        if _launch_info:
            meta.update(_launch_info)
            _launch_info.clear()
        _prof.append((fn.__name__, meta, e0, e1))
        return r
    return wrapper


for _n in ("stem_tail_stats", "stem_tail_pool", "stem_tail_bwd", "stem_tail_pooled", "stem_tail_combine", "conv2d", "conv2d_wgrad", "conv2d_wgrad_fold", "pack_conv_weights", "pack_conv_weights_scaled", "pack_conv_weights_gated", "unpack_conv_wgrad", "pack_bias", "colstats",
           "reduce_partials", "bn_finalize", "bn_apply", "bn_apply_gap", "bn_bwd_reduce", "bn_bwd_finalize", "bn_bwd_apply",
           "maxpool_fwd", "maxpool_bwd", "gap_partial", "gap_finish", "gap_bwd", "eca_gate", "eca_scale",
           "eca_bwd_small", "eca_bwd_apply", "eca_stem_fold", "nchw_to_nhwc", "pad_rows", "gate_mixture_fwd", "gate_mixture_bwd",
           "moe_loss"):
    globals()[_n] = _timed(globals()[_n])


# ---------------------------------------------------------------------------------------------- PU-Net / PMoE
def maxpool2_fwd(x, y, c=None, x_coff=0):
    """MaxPool2d(2,2) of the channel window [x_coff, x_coff+c) of ``x`` into dense ``y`` [N,H/2,W/2,c]."""
    n, h, w_, ld = _nhwc(x, "x")
    c = ld if c is None else c
    if tuple(y.shape) != (n, h // 2, w_ // 2, c):
        raise ValueError(f"maxpool2_fwd: y must be {(n, h // 2, w_ // 2, c)}, got {tuple(y.shape)}")
    check(load().pmoe_maxpool2s2_fwd(ptr(x, "x"), ptr(y, "y", x.dtype), n, h, w_, c, ld, x_coff, dt(x), stream_ptr()),
          "pmoe_maxpool2s2_fwd")


def pixel_shuffle2(src, dst, c, dst_coff=0):
    """src [N,H,W,>=4c] (channel (dy*2+dx)*c + k) -> dst[n, 2y+dy, 2x+dx, dst_coff + k]."""
    n, h, w_, sld = _nhwc(src, "src")
    n2, h2, w2, dld = _nhwc(dst, "dst")
    if (n2, h2, w2) != (n, 2 * h, 2 * w_):
        raise ValueError(f"pixel_shuffle2: dst must be [{n},{2 * h},{2 * w_},*], got {tuple(dst.shape)}")
    check(load().pmoe_pixel_shuffle2(ptr(src, "src"), ptr(dst, "dst", src.dtype), n, h, w_, c, sld, dld, dst_coff, dt(src),
                                     stream_ptr()), "pmoe_pixel_shuffle2")


def copy_window(src, src_coff, dst, dst_coff, c):
    """dst[..., dst_coff:dst_coff+c] = src[..., src_coff:src_coff+c] over all leading rows."""
    rows = src.numel() // src.shape[-1]
    if dst.numel() // dst.shape[-1] != rows:
        raise ValueError("copy_window: row counts differ")
    check(load().pmoe_copy_window(ptr(src, "src"), src.shape[-1], src_coff, ptr(dst, "dst", src.dtype), dst.shape[-1],
                                  dst_coff, rows, c, dt(src), stream_ptr()), "pmoe_copy_window")


def maxpool2_bwd(x, dy, dx, dskip=None, c=None, x_coff=0, dskip_coff=0):
    """Backward of MaxPool2d(2,2) over the channel window [x_coff, x_coff+c) of ``x``: ``dx`` (dense [N,H,W,c]) = dy at
    the first maximum of every 2x2 window + the same window of ``dskip`` (skip-concatenation gradient) if given."""
    n, h, w_, ld = _nhwc(x, "x")
    c = ld if c is None else c
    if tuple(dx.shape) != (n, h, w_, c) or tuple(dy.shape) != (n, h // 2, w_ // 2, c):
        raise ValueError(f"maxpool2_bwd: dx must be {(n, h, w_, c)} and dy {(n, h // 2, w_ // 2, c)}")
    if dskip is not None and tuple(dskip.shape[:3]) != (n, h, w_):
        raise ValueError("maxpool2_bwd: dskip must share x's geometry")
    check(load().pmoe_maxpool2s2_bwd(ptr(x, "x"), ld, x_coff, ptr(dy, "dy", x.dtype), ptr(dskip, "dskip", x.dtype),
                                     dskip.shape[-1] if dskip is not None else 0, dskip_coff, ptr(dx, "dx", x.dtype), n, h,
                                     w_, c, dt(x), stream_ptr()), "pmoe_maxpool2s2_bwd")


def pixel_unshuffle2(src, dst, c, src_coff=0):
    """dst[n,y,x,(dy*2+dx)*c + k] = src[n,2y+dy,2x+dx,src_coff + k]  (backward of pixel_shuffle2)."""
    n, h2, w2, sld = _nhwc(src, "src")
    if tuple(dst.shape[:3]) != (n, h2 // 2, w2 // 2) or dst.shape[-1] < 4 * c:
        raise ValueError("pixel_unshuffle2: dst must be [N,H/2,W/2,>=4c]")
    check(load().pmoe_pixel_unshuffle2(ptr(src, "src"), sld, src_coff, ptr(dst, "dst", src.dtype), dst.shape[-1], n, h2 // 2,
                                       w2 // 2, c, dt(src), stream_ptr()), "pmoe_pixel_unshuffle2")


def add_window(src, src_coff, dst, dst_coff, c):
    """dst[..., dst_coff:dst_coff+c] += src[..., src_coff:src_coff+c] over all leading rows."""
    rows = src.numel() // src.shape[-1]
    if dst.numel() // dst.shape[-1] != rows:
        raise ValueError("add_window: row counts differ")
    check(load().pmoe_add_window(ptr(src, "src"), src.shape[-1], src_coff, ptr(dst, "dst", src.dtype), dst.shape[-1],
                                 dst_coff, rows, c, dt(src), stream_ptr()), "pmoe_add_window")


def cat_windows(srcs, dst, c, src_coff=0, dst_c=None):
    """dst[..., k*c:(k+1)*c] = srcs[k][..., src_coff:src_coff+c] for all k (len(srcs) <= 8), zeros up to ``dst_c``
    (default: the whole row of ``dst``) -- torch.cat along channels in one launch."""
    k = len(srcs)
    if not 1 <= k <= 8:
        raise ValueError("cat_windows: 1..8 sources")
    s0 = srcs[0]
    rows = s0.numel() // s0.shape[-1]
    for t in srcs:
        if t.shape != s0.shape or t.dtype != dst.dtype:
            raise ValueError("cat_windows: sources must share shape and the destination's dtype")
    if dst.numel() // dst.shape[-1] != rows:
        raise ValueError("cat_windows: row counts differ")
    arr = (C.c_void_p * k)(*[ptr(t, "src").value for t in srcs])
    check(load().pmoe_cat_windows(arr, k, c, s0.shape[-1], src_coff, ptr(dst, "dst"), dst.shape[-1],
                                  dst.shape[-1] if dst_c is None else dst_c, rows, dt(dst), stream_ptr()), "pmoe_cat_windows")


def nhwc_to_nchw(src, dst, c, src_coff=0):
    """src [N,H,W,ld] (bf16/f32) channel window -> dst f32 [N,c,H,W]."""
    n, h, w_, ld = _nhwc(src, "src")
    if tuple(dst.shape) != (n, c, h, w_):
        raise ValueError(f"nhwc_to_nchw: dst must be {(n, c, h, w_)}, got {tuple(dst.shape)}")
    check(load().pmoe_nhwc_to_nchw(ptr(src, "src"), ld, src_coff, ptr(dst, "dst", torch.float32), n, h * w_, c, dt(src),
                                   stream_ptr()), "pmoe_nhwc_to_nchw")


SEG_MODES = {"tversky": 0, "l1": 1, "l2": 2}


def seg_loss_fwd(logits, target, mode, ce_weight=0.5, tversky_weight=0.5, alpha=0.5, beta=0.5):
    """AutoregressiveCriterion forward.  logits f32 [B,F,C,H,W], target int64 [B,F,H,W] -> (loss[1+F], coefG, coefT)."""
    if logits.dim() != 5 or tuple(target.shape) != (logits.shape[0], logits.shape[1]) + tuple(logits.shape[3:]):
        raise ValueError(f"seg_loss: logits [B,F,C,H,W] and target [B,F,H,W] expected, got {tuple(logits.shape)} / "
                         f"{tuple(target.shape)}")
    b, f, c, h, w_ = logits.shape
    f32, dev = torch.float32, logits.device
    lib = load()
    rows, cp = lib.pmoe_seg_loss_rows(b, h, w_), lib.pmoe_seg_loss_cp(c)
    if rows < 1 or cp < 1:
        raise ValueError(f"seg_loss: unsupported shape (C={c} must be 2..32)")
    m = SEG_MODES[mode]
    partG = torch.empty(f, rows, 4, 32, dtype=f32, device=dev)
    loss = torch.empty(1 + f, dtype=f32, device=dev)
    partT = coefG = coefT = None
    if m == 0:
        partT = torch.empty(f, rows, 3, cp, w_, dtype=f32, device=dev)
        coefG = torch.empty(f, 32, dtype=f32, device=dev)
        coefT = torch.empty(f, 2, cp, w_, dtype=f32, device=dev)
    check(lib.pmoe_seg_loss_fwd(ptr(logits, "logits", f32), ptr(target, "target", torch.int64), b, f, c, h, w_, m, ce_weight,
                                tversky_weight, alpha, beta, ptr(partT), ptr(partG), ptr(coefG), ptr(coefT), ptr(loss),
                                stream_ptr()), "pmoe_seg_loss_fwd")
    return loss, coefG, coefT


def seg_loss_bwd(logits, target, coefG, coefT, dloss, dlogits, mode):
    b, f, c, h, w_ = logits.shape
    f32 = torch.float32
    check(load().pmoe_seg_loss_bwd(ptr(logits, "logits", f32), ptr(target, "target", torch.int64), ptr(coefG), ptr(coefT),
                                   ptr(dloss, "dloss", f32), ptr(dlogits, "dlogits", f32), b, f, c, h, w_, SEG_MODES[mode],
                                   stream_ptr()), "pmoe_seg_loss_bwd")


def action_head_fwd(head, spd, actions, speeds, B):
    f32 = torch.float32
    check(load().pmoe_action_head_fwd(ptr(head, "head"), head.shape[-1], ptr(spd, "spd", head.dtype), spd.shape[-1],
                                      ptr(actions, "actions", f32), ptr(speeds, "speeds", f32), B, dt(head), stream_ptr()),
          "pmoe_action_head_fwd")


def action_head_bwd(actions, dactions, dspeeds, dhead, dspd, B):
    f32 = torch.float32
    check(load().pmoe_action_head_bwd(ptr(actions, "actions", f32),
                                      ptr(dactions, "dactions", f32) if dactions is not None else None,
                                      ptr(dspeeds, "dspeeds", f32) if dspeeds is not None else None,
                                      ptr(dhead, "dhead"), dhead.shape[-1], ptr(dspd, "dspd", dhead.dtype), dspd.shape[-1],
                                      B, dt(dhead), stream_ptr()), "pmoe_action_head_bwd")


def action_loss(actions, speeds, actions_gt, speed_gt, c0, c1, loss, dactions, dspeeds, B):
    f32 = torch.float32
    o = lambda t, n: ptr(t, n, f32) if t is not None else None   # noqa: E731
    check(load().pmoe_action_loss(ptr(actions, "actions", f32), o(speeds, "speeds"), ptr(actions_gt, "actions_gt", f32),
                                  o(speed_gt, "speed_gt"), float(c0), float(c1), ptr(loss, "loss", f32),
                                  ptr(dactions, "dactions", f32), o(dspeeds, "dspeeds"), B, stream_ptr()), "pmoe_action_loss")


def blend_fwd(moe_act, pu_act, lat_w, lat_b, long_w, long_b, out, B):
    f32 = torch.float32
    check(load().pmoe_blend_fwd(ptr(moe_act, "moe_actions", f32), ptr(pu_act, "punet_actions", f32), ptr(lat_w, "lat_w", f32),
                                ptr(lat_b, "lat_b", f32), ptr(long_w, "long_w", f32), ptr(long_b, "long_b", f32),
                                ptr(out, "out", f32), B, stream_ptr()), "pmoe_blend_fwd")


def blend_bwd(moe_act, pu_act, lat_w, long_w, out, dout, dlat_w, dlat_b, dlong_w, dlong_b, dpu, B):
    f32 = torch.float32
    check(load().pmoe_blend_bwd(ptr(moe_act, "moe_actions", f32), ptr(pu_act, "punet_actions", f32), ptr(lat_w, "lat_w", f32),
                                ptr(long_w, "long_w", f32), ptr(out, "out", f32), ptr(dout, "dout", f32),
                                ptr(dlat_w, "dlat_w", f32), ptr(dlat_b, "dlat_b", f32), ptr(dlong_w, "dlong_w", f32),
                                ptr(dlong_b, "dlong_b", f32), ptr(dpu, "dpunet", f32) if dpu is not None else None, B,
                                stream_ptr()), "pmoe_blend_bwd")


for _n in ("maxpool2_fwd", "pixel_shuffle2", "copy_window", "action_head_fwd", "action_head_bwd", "action_loss",
           "blend_fwd", "blend_bwd", "maxpool2_bwd", "pixel_unshuffle2", "add_window", "nhwc_to_nchw", "seg_loss_fwd",
           "seg_loss_bwd", "cat_windows"):
    globals()[_n] = _timed(globals()[_n])
