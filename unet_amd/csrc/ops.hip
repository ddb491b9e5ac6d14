// Operator-level C ABI: single convolution / norm launches with caller-provided tensors.
// Thin wrappers that build the same parameter blocks the network engine uses.
#include <math.h>

#include "common.h"
#include "geom.h"
#include "kparams.h"
#include "../../include/unet_mi355.h"

namespace {

ConvShape to_shape(const mi_conv_desc* d) {
    ConvShape s;
    s.ndim = d->ndim; s.transposed = d->transposed; s.ksize = d->ksize; s.stride = d->stride;
    s.cin = d->cin; s.cout = d->cout; s.nb = d->nb;
    for (int i = 0; i < 3; ++i) s.in_dims[i] = d->in_dims[i];
    return s;
}

int check_desc(const mi_conv_desc* d) {
    MI_CHECK(d, "null descriptor");
    MI_CHECK(d->ndim == 2 || d->ndim == 3, "ndim must be 2 or 3");
    MI_CHECK(d->ksize == 3 || d->ksize == 1, "ksize must be 3 or 1");
    MI_CHECK(d->stride == 1 || d->stride == 2, "stride must be 1 or 2");
    MI_CHECK(!d->transposed || (d->ksize == 3 && d->stride == 2), "transposed convolution is k3 s2 only");
    MI_CHECK(d->ksize == 3 || d->stride == 1, "1x1 convolution is stride 1 only");
    MI_CHECK(d->cin % 16 == 0 && d->cout % 16 == 0 && d->cin > 0 && d->cout > 0, "channels must be multiples of 16");
    MI_CHECK(d->dtype == 0 || d->dtype == 1, "dtype must be 0 or 1");
    MI_CHECK(d->nb > 0, "batch must be positive");
    MI_CHECK(d->ndim == 3 || d->in_dims[0] == 1, "in_dims[0] must be 1 in 2-D");
    return 0;
}

}  // namespace

extern "C" {

size_t mi_conv_packed_bytes(const mi_conv_desc* d) {
    if (!d) return 0;
    const ConvShape s = to_shape(d);
    return (size_t)ktaps(s) * d->cin * d->cout * (d->dtype ? 2 : 4);
}

int mi_conv_pack(const mi_conv_desc* d, int dgrad, const float* w, void* packed, void* stream) {
    if (int r = check_desc(d)) return r;
    MI_CHECK(w && packed, "null argument");
    const ConvShape s = to_shape(d);
    IgemmParams g; memset(&g, 0, sizeof(g));
    int kidx[MI_MAX_TAPS], ntaps = 0;
    MI_CHECK(igemm_geometry(s, dgrad ? DIR_DGRAD : DIR_FWD, g, kidx, &ntaps), "odd size under stride 2");
    const int KK = ktaps(s);
    PackDesc p; memset(&p, 0, sizeof(p));
    p.src = w; p.dst = packed; p.ntaps = ntaps;
    for (int i = 0; i < ntaps; ++i) p.kidx[i] = (signed char)kidx[i];
    // element (co, ci, k) of the torch tensor: conv [co][ci][k], transposed conv [ci][co][k]
    const int s_co = d->transposed ? KK : d->cin * KK;
    const int s_ci = d->transposed ? d->cout * KK : KK;
    if (!dgrad) { p.N = d->cout; p.K = d->cin; p.sn = s_co; p.sk = s_ci; }
    else { p.N = d->cin; p.K = d->cout; p.sn = s_ci; p.sk = s_co; }
    p.Ntot = p.N; p.Ktot = p.K;
    return launch_pack_one(d->dtype, p, (hipStream_t)stream);
}

static int run_igemm(const mi_conv_desc* d, int dir, const mi_tensor* in, const void* packed, const float* bias,
                     const mi_tensor* out, const mi_tensor* addend, double* stats, void* stream) {
    if (int r = check_desc(d)) return r;
    MI_CHECK(in && in->ptr && packed && out && out->ptr, "null argument");
    const ConvShape s = to_shape(d);
    IgemmParams p; memset(&p, 0, sizeof(p));
    int kidx[MI_MAX_TAPS], ntaps = 0;
    MI_CHECK(igemm_geometry(s, dir, p, kidx, &ntaps), "odd size under stride 2");
    p.K = dir == DIR_FWD ? d->cin : d->cout;
    p.N = dir == DIR_FWD ? d->cout : d->cin;
    p.in0 = (const char*)in->ptr; p.in0_pitch = in->pitch; p.in0_coff = in->coff; p.in0_c = p.K;
    if (in->ptr2) { p.in1 = (const char*)in->ptr2; p.in1_pitch = in->pitch2; p.in1_coff = in->coff2; p.in0_c = in->c1; }
    p.w = (const char*)packed; p.bias = bias; p.bias_c0 = 0;
    p.out = (char*)out->ptr; p.out_pitch = out->pitch; p.out_coff = out->coff;
    if (addend && addend->ptr) { p.addend = (const char*)addend->ptr; p.add_pitch = addend->pitch; p.add_coff = addend->coff; }
    if (stats) { p.stats = stats; p.stats_c = p.N; p.nstat = p.N; }
    return launch_igemm(d->dtype, p, (hipStream_t)stream);
}

int mi_conv_forward(const mi_conv_desc* d, const mi_tensor* x, const void* packed, const float* bias, const mi_tensor* y,
                    const mi_tensor* addend, double* stats, void* stream) {
    return run_igemm(d, DIR_FWD, x, packed, bias, y, addend, stats, stream);
}

int mi_conv_dgrad(const mi_conv_desc* d, const mi_tensor* dy, const void* packed_dgrad, const mi_tensor* dx,
                  const mi_tensor* addend, void* stream) {
    return run_igemm(d, DIR_DGRAD, dy, packed_dgrad, nullptr, dx, addend, nullptr, stream);
}

static void wgrad_params(const mi_conv_desc* d, WgradParams& p) {
    const ConvShape s = to_shape(d);
    memset(&p, 0, sizeof(p));
    wgrad_geometry(s, d->dtype ? 2 : 4, p);
    p.F = d->transposed ? d->cin : d->cout;
    p.S = d->transposed ? d->cout : d->cin;
    p.f0_c = p.F;
    wgrad_choose_split(d->dtype, p);
}

size_t mi_conv_wgrad_workspace_bytes(const mi_conv_desc* d) {
    if (!d) return 0;
    WgradParams p;
    wgrad_params(d, p);
    const size_t slabs = wgrad_slabs(d->dtype, p);
    return slabs * ((size_t)p.ntaps_total * p.F * p.S + p.F) * sizeof(float) + 256;
}

int mi_conv_wgrad(const mi_conv_desc* d, const mi_tensor* x, const mi_tensor* dy, float* dw, float* db, void* workspace,
                  void* stream) {
    if (int r = check_desc(d)) return r;
    MI_CHECK(x && x->ptr && dy && dy->ptr && dw && workspace, "null argument");
    WgradParams p;
    wgrad_params(d, p);
    const mi_tensor* f = d->transposed ? x : dy;
    const mi_tensor* sh = d->transposed ? dy : x;
    p.f0 = (const char*)f->ptr; p.f0_pitch = f->pitch; p.f0_coff = f->coff;
    if (f->ptr2) { p.f1 = (const char*)f->ptr2; p.f1_pitch = f->pitch2; p.f1_coff = f->coff2; p.f0_c = f->c1; }
    p.s = (const char*)sh->ptr; p.s_pitch = sh->pitch; p.s_coff = sh->coff;
    const size_t slabs = wgrad_slabs(d->dtype, p);
    p.partial = (float*)workspace;
    p.partial_bias = p.partial + slabs * (size_t)p.ntaps_total * p.F * p.S;
    if (int r = launch_wgrad(d->dtype, p, (hipStream_t)stream)) return r;
    ReduceDesc rd; memset(&rd, 0, sizeof(rd));
    rd.partial = p.partial; rd.partial_bias = p.partial_bias; rd.nslab = (int)slabs; rd.ntaps = p.ntaps_total;
    rd.Ftot = p.F; rd.S = p.S; rd.f_off = 0; rd.Fcnt = p.F; rd.KK = p.ntaps_total;
    for (int i = 0; i < p.ntaps_total; ++i) rd.kidx[i] = (signed char)i;
    rd.dst_w = dw; rd.dst_b = (db && !d->transposed) ? db : nullptr;
    return launch_reduce_one(rd, (hipStream_t)stream);
}

static int conv1_geometry(int ndim, int nb, const int32_t* fine_dims, int n, C1Params& p) {
    MI_CHECK(ndim == 2 || ndim == 3, "ndim must be 2 or 3");
    MI_CHECK(nb > 0 && fine_dims, "batch must be positive, dims must be given");
    MI_CHECK(n % 8 == 0 && n >= 8 && n <= 512, "channels must be a multiple of 8, at most 512");
    MI_CHECK(ndim == 3 || fine_dims[0] == 1, "fine_dims[0] must be 1 in 2-D");
    memset(&p, 0, sizeof(p));
    p.ndim = ndim; p.nb = nb; p.N = n;
    for (int i = 0; i < 3; ++i) {
        const bool flat = ndim == 2 && i == 0;
        MI_CHECK(fine_dims[i] > 0 && (flat || fine_dims[i] % 2 == 0), "odd size under stride 2");
        p.fd[i] = fine_dims[i]; p.cd[i] = flat ? 1 : fine_dims[i] / 2;
    }
    p.nsplit = c1_wgrad_slabs(n, p.cd);
    return 0;
}

size_t mi_conv1_wgrad_workspace_bytes(int ndim, int n, const int32_t* fine_dims) {
    C1Params p;
    if (conv1_geometry(ndim, 1, fine_dims, n, p)) return 0;
    return (size_t)p.nsplit * ((ndim == 3 ? 27 : 9) + 1) * n * sizeof(float) + 256;
}

int mi_conv1_wgrad(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const float* x, const mi_tensor* y, float* dw,
                   float* db, void* workspace, void* stream) {
    MI_CHECK(dtype == 0 || dtype == 1, "dtype must be 0 or 1");
    MI_CHECK(x && y && y->ptr && dw && workspace, "null argument");
    C1Params p;
    if (int r = conv1_geometry(ndim, nb, fine_dims, n, p)) return r;
    const int KK = ndim == 3 ? 27 : 9;
    p.x1 = x;
    p.y = (const char*)y->ptr; p.y_pitch = y->pitch; p.y_coff = y->coff; p.n_y1 = n;
    if (y->ptr2) { p.y2 = (const char*)y->ptr2; p.y2_pitch = y->pitch2; p.y2_coff = y->coff2; p.n_y1 = y->c1; }
    p.partial = (float*)workspace;
    p.partial_bias = p.partial + (size_t)p.nsplit * KK * n;
    if (int r = launch_c1_s2_wgrad(dtype, p, (hipStream_t)stream)) return r;
    ReduceDesc rd; memset(&rd, 0, sizeof(rd));
    rd.partial = p.partial; rd.partial_bias = p.partial_bias; rd.nslab = p.nsplit; rd.ntaps = KK;
    rd.Ftot = n; rd.S = 1; rd.f_off = 0; rd.Fcnt = n; rd.KK = KK;
    for (int i = 0; i < KK; ++i) rd.kidx[i] = (signed char)i;
    rd.dst_w = dw; rd.dst_b = db;
    return launch_reduce_one(rd, (hipStream_t)stream);
}

int mi_conv1_forward(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const float* x, const float* w,
                     const float* bias, const mi_tensor* y, double* stats, void* stream) {
    MI_CHECK(dtype == 0 || dtype == 1, "dtype must be 0 or 1");
    MI_CHECK(x && w && y && y->ptr, "null argument");
    C1Params p;
    if (int r = conv1_geometry(ndim, nb, fine_dims, n, p)) return r;
    p.x1 = x; p.w = w; p.n_w1 = n; p.bias = bias; p.bias_c0 = 0;
    p.y_out = (char*)y->ptr; p.y_pitch = y->pitch; p.y_coff = y->coff; p.n_o1 = n;
    if (y->ptr2) { p.y2_out = (char*)y->ptr2; p.y2o_pitch = y->pitch2; p.y2o_coff = y->coff2; p.n_o1 = y->c1; }
    if (stats) { p.stats = stats; p.stats_c = n; p.nstat = n; }
    return launch_c1_s2_fwd(dtype, p, (hipStream_t)stream);
}

int mi_conv1_transposed(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const mi_tensor* y, const float* w,
                        const float* bias, float* x_out, double* stats, void* stream) {
    MI_CHECK(dtype == 0 || dtype == 1, "dtype must be 0 or 1");
    MI_CHECK(y && y->ptr && w && x_out, "null argument");
    C1Params p;
    if (int r = conv1_geometry(ndim, nb, fine_dims, n, p)) return r;
    p.x1_out = x_out; p.w = w; p.bias = bias;
    p.y = (const char*)y->ptr; p.y_pitch = y->pitch; p.y_coff = y->coff; p.n_y1 = n;
    if (y->ptr2) { p.y2 = (const char*)y->ptr2; p.y2_pitch = y->pitch2; p.y2_coff = y->coff2; p.n_y1 = y->c1; }
    if (stats) { p.stats = stats; p.stats_c = 1; p.nstat = 1; }
    return launch_c1_s2_T(dtype, p, (hipStream_t)stream);
}

static void fill_drop(NormParams& p, float dropout, uint64_t seed, uint32_t layer) {
    p.seed = seed; p.layer = layer;
    norm_set_dropout(p, dropout);
}

int mi_instnorm_prelu_fwd(int dtype, int nb, int64_t npix, int c, const mi_tensor* raw, const double* stats,
                          const float* alpha, const mi_tensor* res, const mi_tensor* out, float dropout, uint64_t seed,
                          uint32_t layer, void* stream) {
    MI_CHECK(raw && raw->ptr && stats && alpha && out && out->ptr, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL, "bad pixel count");
    NormParams p; memset(&p, 0, sizeof(p));
    p.x = (const char*)raw->ptr; p.x_pitch = raw->pitch; p.x_coff = raw->coff;
    if (res && res->ptr) { p.res = (const char*)res->ptr; p.res_pitch = res->pitch; p.res_coff = res->coff; }
    p.out = (char*)out->ptr; p.out_pitch = out->pitch; p.out_coff = out->coff;
    p.stats = stats; p.alpha = alpha; p.nb = nb; p.npix = (int)npix; p.c = c; p.eps = 1e-5f;
    fill_drop(p, dropout, seed, layer);
    return launch_norm_fwd(dtype, p, (hipStream_t)stream);
}

int mi_instnorm_prelu_bwd(int dtype, int nb, int64_t npix, int c, const mi_tensor* dy, const mi_tensor* raw,
                          const double* stats, const float* alpha, double* bstats, float* dalpha, const mi_tensor* dx,
                          float dropout, uint64_t seed, uint32_t layer, void* stream) {
    MI_CHECK(raw && raw->ptr && dy && dy->ptr && stats && alpha && bstats && dx && dx->ptr, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL, "bad pixel count");
    NormParams p; memset(&p, 0, sizeof(p));
    p.x = (const char*)raw->ptr; p.x_pitch = raw->pitch; p.x_coff = raw->coff;
    p.dy = (const char*)dy->ptr; p.dy_pitch = dy->pitch; p.dy_coff = dy->coff;
    p.out = (char*)dx->ptr; p.out_pitch = dx->pitch; p.out_coff = dx->coff;
    p.stats = stats; p.bstats = bstats; p.alpha = alpha; p.dalpha = dalpha;
    p.nb = nb; p.npix = (int)npix; p.c = c; p.eps = 1e-5f;
    fill_drop(p, dropout, seed, layer);
    MI_HIP(hipMemsetAsync(bstats, 0, (size_t)nb * MI_NSLOT * c * 2 * sizeof(double), (hipStream_t)stream));
    return launch_norm_bwd(dtype, p, (hipStream_t)stream);
}

}  // extern "C"
