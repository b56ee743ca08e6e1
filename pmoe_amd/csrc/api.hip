// C-ABI entry points for the MFMA conv / grouped-GEMM kernels, plus version and error strings.
#include "common.h"
#include "kernels.h"

static ConvArgs to_args(const pmoe_conv_desc* d) {
    ConvArgs a;
    a.in = d->in; a.w = d->w; a.out = d->out; a.res = d->res; a.bias = d->bias; a.stats = d->stats;
    a.N = d->n; a.H = d->h; a.W = d->w_; a.Cin = d->cin;
    a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout; a.CoutP = d->coutp;
    a.in_ld = d->in_ld; a.in_coff = d->in_coff; a.out_ld = d->out_ld; a.out_coff = d->out_coff;
    a.res_ld = d->res_ld; a.res_coff = d->res_coff;
    a.ipe = d->ipe; a.in_shared = d->in_shared;
    a.ks = d->ks; a.stride = d->stride; a.pad = d->pad; a.dilate = d->dilate;
    a.act = d->act; a.res_mode = (d->res || d->res_mode == PMOE_RES_INBN) ? d->res_mode : PMOE_RES_NONE;
    a.drop_p = d->drop_p; a.seed = d->seed;
    a.oscale = d->out_scale; a.in_scale = d->in_scale; a.w_fp8 = d->w_fp8; a.in_fp8 = d->in_fp8;
    a.bn = d->bn_coef; a.bn_ipe = d->bn_ipe > 0 ? d->bn_ipe : d->ipe;
    a.lTW = a.lTH = a.TN = a.n_groups = a.tiles_y = a.tiles_x = 0;
    a.kh = a.kw = d->ks; a.use_tapmap = 0; a.tapmap[0] = a.tapmap[1] = a.tapmap[2] = a.tapmap[3] = 0;
    a.out_step = 1; a.out_offy = a.out_offx = 0; a.OH = d->ho; a.OW = d->wo; a.prefetch = 0; a.stagger = 0;
    a.shuf_c = d->shuffle_c > 0 ? d->shuffle_c : 0;
    return a;
}

static int check_conv(const pmoe_conv_desc* d) {
    if (!d || !d->in || !d->w || !d->out) return PMOE_ERR_ARG;
    if (d->n <= 0 || d->h <= 0 || d->w_ <= 0 || d->ho <= 0 || d->wo <= 0 || d->ipe <= 0) return PMOE_ERR_ARG;
    const int ve = d->dtype == PMOE_DT_BF16 ? 8 : 4;
    if (d->in_fp8 && (!d->w_fp8 || d->dtype != PMOE_DT_BF16 || d->in_ld % 16 || d->in_coff % 16)) return PMOE_ERR_ARG;
    if (d->in_ld % ve || d->in_coff % ve || d->out_ld % ve || d->out_coff % ve) return PMOE_ERR_ARG;
    if (d->res && (d->res_ld % ve || d->res_coff % ve)) return PMOE_ERR_ARG;
    if (d->shuffle_c > 0 ? (d->cout != 4 * d->shuffle_c || d->shuffle_c % ve || d->out_coff + d->shuffle_c > d->out_ld || d->cout > d->coutp)
                         : (d->cout > d->coutp || d->out_coff + d->cout > d->out_ld)) return PMOE_ERR_ARG;
    if (d->in_coff + d->cin > d->in_ld) return PMOE_ERR_ARG;
    if (d->drop_p < 0.f || d->drop_p >= 1.f) return PMOE_ERR_ARG;
    if (d->res && d->res_mode == PMOE_RES_DBN &&
        (!d->bn_coef || !d->stats || d->dtype != PMOE_DT_BF16 || (d->bn_ipe > 0 && (d->bn_ipe % d->ipe || d->n % d->bn_ipe))))
        return PMOE_ERR_ARG;
    if (d->res_mode == PMOE_RES_INBN &&
        (d->res || !d->bn_coef || d->dtype != PMOE_DT_BF16 || (d->bias && d->ks != 1) || d->act != PMOE_ACT_NONE || d->drop_p != 0.f || d->in_shared ||
         d->w_fp8 || d->dilate || (d->bn_ipe > 0 && (d->bn_ipe % d->ipe || d->n % d->bn_ipe))))
        return PMOE_ERR_ARG;
    // geometry: forward conv / transposed (dilate) relation between (h,w) and (ho,wo)
    if (!d->dilate) {
        if (d->ho != (d->h + 2 * d->pad - d->ks) / d->stride + 1) return PMOE_ERR_ARG;
        if (d->wo != (d->w_ + 2 * d->pad - d->ks) / d->stride + 1) return PMOE_ERR_ARG;
    }
    return 0;
}

extern "C" {

int pmoe_version(void) { return PMOE_ABI_VERSION; }

int pmoe_abi_sizeof(int which) {
    return which == 0 ? (int)sizeof(pmoe_conv_desc) : which == 1 ? (int)sizeof(pmoe_wgrad_desc)
           : which == 2 ? (int)sizeof(pmoe_opt_tensor) : PMOE_ERR_ARG;
}

const char* pmoe_error_string(int code) {
    if (code == 0) return "ok";
    if (code == PMOE_ERR_ARG) return "pmoe: invalid argument (shape / alignment / pointer)";
    if (code == PMOE_ERR_UNSUPPORTED) return "pmoe: unsupported configuration (no tile fits LDS)";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "pmoe: unknown error";
}

int pmoe_conv2d_igemm(const pmoe_conv_desc* d, void* stream) {
    int rc = check_conv(d);
    if (rc) return rc;
    return conv_igemm_launch(to_args(d), d->dtype, (hipStream_t)stream);
}

// planning descriptors may leave the leading dimensions 0 = dense rows (cin / cout wide): the kernel choice depends on them
// (32-bit offset ranges), so the launch and the planning call must see the same values
static ConvArgs to_plan_args(const pmoe_conv_desc* d) {
    ConvArgs a = to_args(d);
    if (a.in_ld <= 0) a.in_ld = a.in_coff + a.Cin;
    if (a.out_ld <= 0) a.out_ld = a.out_coff + a.Cout;
    return a;
}

int pmoe_conv2d_stat_rows(const pmoe_conv_desc* d) {
    if (!d || d->ipe <= 0) return PMOE_ERR_ARG;
    return conv_igemm_mblocks(to_plan_args(d), d->dtype);
}

int pmoe_conv2d_plan(const pmoe_conv_desc* d) {
    if (!d || d->ipe <= 0) return PMOE_ERR_ARG;
    return conv_igemm_plan(to_plan_args(d), d->dtype);
}

static int to_wgrad_args(const pmoe_wgrad_desc* d, WgradArgs& a) {
    if (!d || d->ipe <= 0) return PMOE_ERR_ARG;
    const int ve = d->dtype == PMOE_DT_BF16 ? 8 : 4;
    if (d->x_ld % ve || d->x_coff % ve || d->dy_ld % ve || d->dy_coff % ve) return PMOE_ERR_ARG;
    if (d->x_coff + d->cin > d->x_ld || d->dy_coff + d->cout > d->dy_ld) return PMOE_ERR_ARG;
    if (d->ho != (d->h + 2 * d->pad - d->ks) / d->stride + 1) return PMOE_ERR_ARG;
    if (d->wo != (d->w_ + 2 * d->pad - d->ks) / d->stride + 1) return PMOE_ERR_ARG;
    a.x = d->x; a.dy = d->dy; a.dw = d->dw_ws; a.part = d->part_ws; a.part_floats = d->part_ws_floats;
    a.N = d->n; a.H = d->h; a.W = d->w_; a.Cin = d->cin; a.CinP = d->cinp;
    a.Ho = d->ho; a.Wo = d->wo; a.Cout = d->cout; a.CoutP = d->coutp;
    a.x_ld = d->x_ld; a.x_coff = d->x_coff; a.dy_ld = d->dy_ld; a.dy_coff = d->dy_coff;
    a.ipe = d->ipe; a.x_shared = d->x_shared;
    a.ks = d->ks; a.stride = d->stride; a.pad = d->pad; a.per_image = d->per_image;
    a.grads = d->grads; a.cout_real = d->cout_real; a.cin_real = d->cin_real; a.defer_fold = d->defer_fold;
    if (a.grads && (a.per_image || a.cout_real <= 0 || a.cin_real <= 0 || a.cout_real > a.CoutP || a.cin_real > a.CinP)) return PMOE_ERR_ARG;
    a.lTW = a.lTH = a.TN = a.n_groups = a.tiles_y = a.tiles_x = a.mb_per_wg = 0;
    a.slice_fastest = 0;
    return 0;
}

int pmoe_conv2d_wgrad(const pmoe_wgrad_desc* d, void* stream) {
    WgradArgs a;
    const int rc = to_wgrad_args(d, a);
    if (rc) return rc;
    if (!d->x || !d->dy || !d->dw_ws) return PMOE_ERR_ARG;
    if (d->bn_fused)
        return conv_wgrad_bnbwd_launch(a, d->bn_z, d->bn_z_ld, d->bn_coef, d->bn_c1, d->bn_c2, d->dtype, (hipStream_t)stream, false);
    return conv_wgrad_launch(a, d->dtype, (hipStream_t)stream);
}

int pmoe_conv2d_wgrad_fold(const pmoe_wgrad_desc* d, void* stream) {
    WgradArgs a;
    const int rc = to_wgrad_args(d, a);
    if (rc) return rc;
    if (!d->dw_ws) return PMOE_ERR_ARG;
    return conv_wgrad_fold(a, d->dtype, (hipStream_t)stream);
}

int pmoe_conv2d_wgrad_plan(const pmoe_wgrad_desc* d) {
    WgradArgs a;
    const int rc = to_wgrad_args(d, a);
    if (rc) return rc;
    if (d->bn_fused) {
        const int r2 = conv_wgrad_bnbwd_launch(a, nullptr, d->bn_z_ld, nullptr, nullptr, nullptr, d->dtype, nullptr, true);
        return r2 ? r2 : 7209;
    }
    return conv_wgrad_plan(a, d->dtype);
}

int64_t pmoe_conv2d_wgrad_ws_floats(const pmoe_wgrad_desc* d) {
    WgradArgs a;
    const int rc = to_wgrad_args(d, a);
    if (rc) return rc;
    if (d->bn_fused) return 0;                            // per image: no K-split scratch
    return conv_wgrad_ws_floats(a, d->dtype);
}

}  // extern "C"
