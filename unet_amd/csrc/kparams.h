// Kernel parameter blocks (passed by value as kernel arguments) and launcher prototypes.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define MI_MAX_TAPS 27
#define MI_MAX_PHASE 8

// Gather-GEMM ("implicit GEMM") forward / dgrad launch.
//   out[opix(b)][n] = sum_{t in taps(phase)} sum_k A[ipix(b,t)][k] * W[t][n][k]   (+bias, +addend)
// b runs over a "base grid"; ipix(b,t) = b*sa + off_t (zero outside the tensor),
// opix(b) = b*so + phase_off.  Convolution: sa=stride, so=1, one phase.  Transposed
// convolution (and the data gradient of a strided convolution): sa=1, so=2, 2^ndim phases.
struct IgemmParams {
    const char* in0; const char* in1;       // contraction channels [0,in0_c) come from in0, the rest from in1
    int in0_pitch, in0_coff, in0_c, in1_pitch, in1_coff;
    int K;                                  // contraction channels in total
    const char* w;                          // packed weights [tap][N][K], storage type of the activations
    const float* bias; int bias_c0;         // bias[ch-bias_c0] is added to output channels >= bias_c0 (null: none)
    char* out; int out_pitch, out_coff, N;
    char* out1; int out1_pitch, out1_coff, out0_c;  // optional second destination: output channels >= out0_c go to out1 (null: none)
    const char* addend; int add_pitch, add_coff;  // optional tensor added in the epilogue (may alias out)
    double* stats; int stats_c, nstat;      // per-(image,slot,channel) {sum, sum of squares} of the first nstat channels
    int nb;
    int id[3], od[3], bd[3];                // gathered-tensor dims, output dims, base-grid dims (z,y,x)
    int sa, so;
    int tlog[3];                            // log2 of the tile dims; tile = 256 base pixels
    int ntile[3];
    int pd[3];                              // LDS patch dims (pixels)
    int omin[3];                            // smallest gather offset per dim
    int nphase;
    int phase_off[MI_MAX_PHASE][3];
    int phase_start[MI_MAX_PHASE + 1];      // tap range of each phase in tap_off / w
    int tap_off[MI_MAX_TAPS];               // LDS patch pixel offset of each tap
    int tapg;                               // taps staged per weight tile (<= 9)
    unsigned mg_pd2, mg_pd12;               // reciprocals of pd[2] and pd[1]*pd[2], filled in by the launcher
    int ksplit; float* kpart;               // > 1: the contraction chunks are split over ksplit workgroups that write raw fp32
                                            // partial sums [ksplit][output pixel][N] to kpart; launch_ksplit_finish completes the layer
    int kpart_blocked;                      // partial sums as [ksplit][image][N / 8][pixel of the image][8] instead: the layout
                                            // launch_finish_norm reads (a workgroup there owns one image and eight channels)
    int tpw;                                // 2-D specialisation: consecutive tiles per workgroup, filled in by the launcher
    // 3-D gathers with stride 2 (sa = 2, one phase): the tap slots are ordered by polyphase component of the gathered
    // tensor (component c = (rz, ry, rx): pixels 2 i - r per dimension; a tap with k = 1 lives in r = 0 at offset 0,
    // k = 0 and k = 2 in r = 1 at offsets 0 and 1), comp_start[c] .. comp_start[c + 1] are the slots of component c,
    // ctap_off their pixel offsets inside a component patch of cpd pixels (tile + 1 per dimension), in which the
    // gather has unit stride.  tap_off / pd describe the same slots for a kernel that stages the whole strided patch.
    int comp_mode;
    int comp_start[9];
    int ctap_off[MI_MAX_TAPS];
    int cpd[3];
    unsigned* start_flag; unsigned start_val;   // not null: the first workgroup stores start_val there as the kernel starts (see NormParams)
};

// Weight-gradient launch: partial[slab][tap][f][s] = sum_{base pixels of the slab}
//   Fm[b][f] * Sm[b*ss + off_tap][s]; plus partial_bias[slab][f] = sum_b Fm[b][f].
struct WgradParams {
    const char* f0; const char* f1;         // "fixed" operand (at base pixels), channels [0,f0_c) from f0, rest from f1
    int f0_pitch, f0_coff, f0_c, f1_pitch, f1_coff, F;
    const char* s; int s_pitch, s_coff, S;  // "shifted" operand
    int nb;
    int bd[3], sd[3];
    int ss;
    int tlog[3], ntile[3], pd[3], omin[3];
    int ntapg;                              // tap groups (3-D: one per kz plane)
    int tg_zoff[3];                         // extra z offset of the patch origin per tap group
    int tap_off[9];                         // LDS patch pixel offsets of the taps of one group
    int ntaps_total;
    float* partial; float* partial_bias;
    int tiles_total, nsplit;
    unsigned mg_pd2, mg_pd12;               // reciprocals of pd[2] and pd[1]*pd[2], filled in by the launcher
};

// Reduction of weight-gradient slabs into the state_dict layout:
//   dst_w[(f*S + s)*KK + kidx[tap]] = sum_slab partial[slab][tap][f_off+f][s]
struct ReduceDesc {
    const float* partial; const float* partial_bias;
    int nslab, ntaps, Ftot, S, f_off, Fcnt, KK;
    signed char kidx[MI_MAX_TAPS + 1];
    float* dst_w; float* dst_b;             // dst_b may be null
    int blk0, nblk;                         // table launches: this descriptor owns workgroups [blk0, blk0 + nblk) of its launch
    int tiled_max_slabs;                    // LDS-tiled form for at most this many slabs (0: never)
};

// Weight packing: dst[((t*Ntot + n_off + n)*Ktot + k_off + k)] = src[n*sn + k*sk + kidx[t]]
struct PackDesc {
    const float* src; void* dst;
    int N, K, ntaps, sn, sk, Ntot, Ktot, n_off, k_off;
    signed char kidx[MI_MAX_TAPS + 1];
    int blk0, nblk;                         // table launches: this descriptor owns workgroups [blk0, blk0 + nblk) of its launch
};

// Descriptor tables travel as kernel arguments (by value, under 4 KiB): nothing is staged in device memory, so a
// launch never depends on what an earlier call left behind in a workspace.
#define MI_PACK_TAB 40
#define MI_REDUCE_TAB 36
struct PackTable { int n, pad; PackDesc d[MI_PACK_TAB]; };
struct ReduceTable { int n, pad; ReduceDesc d[MI_REDUCE_TAB]; };

// Instance-norm + PReLU (+dropout, +residual) launch
struct NormParams {
    const char* x; int x_pitch, x_coff;     // raw conv output
    const char* dy; int dy_pitch, dy_coff;  // backward: gradient of the activated output
    const char* res; int res_pitch, res_coff;  // forward: optional residual added after the activation
    char* out; int out_pitch, out_coff;     // forward: activated output; backward: gradient of x
    const double* stats;                    // forward statistics [nb][NSLOT][c][2] (fp64 accumulators)
    double* bstats;                         // backward statistics [nb][NSLOT][c][2]
    const float* alpha;                     // PReLU slope (device scalar)
    float* dalpha;                          // gradient of the slope: accumulated scalar, or per-workgroup partials
    int dalpha_slots;                       // 0: atomicAdd into dalpha[0]; 1: dalpha[img*gridDim.x + blockIdx.x] = partial
    int nb, npix, c;
    float eps;
    unsigned long long seed; unsigned layer, drop_thresh; float drop_scale;  // drop_thresh: P(drop) * 65536, 0 = no dropout
    unsigned* start_flag; unsigned start_val;   // backward: not null -> the first workgroup stores start_val there as the kernel starts (every
                                            // earlier kernel of the stream is complete then): releases a side stream that waits on it
};
// dropout probability -> 16-bit threshold and survivor scale of NormParams
inline void norm_set_dropout(NormParams& p, float dropout) {
    if (dropout > 0.f) {
        long t = (long)((double)dropout * 65536.0 + 0.5);
        p.drop_thresh = t < 1 ? 1u : t > 65535 ? 65535u : (unsigned)t;
        p.drop_scale = 1.0f / (1.0f - dropout);
    } else { p.drop_thresh = 0; p.drop_scale = 1.f; }
}



// One-channel <-> N-channel stride-2 kernels (first encoder conv, last decoder transposed conv)
// and the one-channel 3^d stencil (top conv_only residual unit).
//   fine grid  : one-channel fp32 tensor [nb][fd0][fd1][fd2]
//   coarse grid: N-channel channels-last tensor at fd/2
struct C1Params {
    const float* x1; float* x1_out;         // fine one-channel tensor (input / output)
    const char* y; char* y_out;             // coarse N-channel tensor (input / output)
    int y_pitch, y_coff, N;
    const char* y2; int y2_pitch, y2_coff;  // input side: channels >= n_y1 are read from this tensor (n_y1 >= N: single source)
    int n_y1;
    char* y2_out; int y2o_pitch, y2o_coff;  // output side: channels >= n_o1 are written to this tensor (null: single destination)
    int n_o1;
    const float* w; const float* w2; int n_w1;  // weights [n][KK]; channels >= n_w1 read w2[(n-n_w1)][KK]
    const float* bias; int bias_c0;
    double* stats; int stats_c, nstat;
    int ndim, nb;
    int fd[3], cd[3];
    float* partial; float* partial_bias;    // weight-gradient slabs [slab][KK][N], [slab][N]
    int nsplit;
};
struct StencilParams {
    const float* in; const float* dout; float* out;
    const float* w; const float* bias; const float* addend;
    int flip, ndim, nb;
    int d[3];
    float* partial; float* partial_bias;    // [slab][KK], [slab]
    int nsplit;
    // 2-D strip kernel, forward: `in` is a RAW one-channel convolution output; instance norm (statistics nrm_stats,
    // [nb][NSLOT][2] fp64 sums over the image), dropout and PReLU are applied as the rows are loaded (pixels outside
    // the image stay zero), the activated values of the thread's own pixels go to act_out (nullptr: nobody reads
    // them) and serve as the identity addend.
    const double* nrm_stats; const float* nrm_alpha; float nrm_eps;
    unsigned long long nrm_seed; unsigned nrm_layer, nrm_thresh; float nrm_scale;
    float* act_out;
};

#ifdef __cplusplus
extern "C++" {
int launch_igemm(int dtype, const IgemmParams& p, hipStream_t st);
int launch_wgrad(int dtype, const WgradParams& p, hipStream_t st);
bool conv2d_dma_applicable(int dtype, const IgemmParams& p);   // conv2d.hip: asynchronously staged 2-D kernels
int launch_conv2d_dma(int dtype, const IgemmParams& p, hipStream_t st);
bool conv2d_dma_enabled();
bool conv3d_k16_applicable(int dtype, const IgemmParams& p);   // conv2d.hip: 3-D 3 x 3 x 3 layers with 16 bf16 contraction channels
int launch_conv3d_k16(const IgemmParams& p, hipStream_t st);
bool igemm_dma_applicable(int dtype, const IgemmParams& p);   // igemm_dma.hip: asynchronously staged kernel for any tile geometry (unit gather stride)
bool igemm_dma_config(int dtype, const IgemmParams& p, int* cps, int* nfrag);
int launch_igemm_dma(int dtype, const IgemmParams& p, hipStream_t st);
void igemm_dma_label(int dtype, const IgemmParams& p, char* buf, size_t n);
void conv2d_dma_label(int dtype, const IgemmParams& p, char* buf, size_t n);   // step-table name of the kernel that serves p
int launch_wgrad_dma(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, hipStream_t st);   // 1: shape not served
int launch_wgrad_thick(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, hipStream_t st);   // wgrad3.hip; 1: shape not served
bool wgrad_thick_config(int dtype, const WgradParams& p, int* ff, int* sf, int* wf, int* wsn);   // block shape wgrad3.hip wants for p
void wgrad_thick_label(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, char* buf, size_t n);   // its kernel's name, if it serves p
int launch_reduce(const ReduceTable& tab, int total_blocks, hipStream_t st);   // total_blocks: sum of the descriptors' nblk
int reduce_blocks_for(int total);
int launch_pack(int dtype, const PackTable& tab, int total_blocks, hipStream_t st);
int pack_blocks_for(int pairs);
int launch_reduce_one(const ReduceDesc& d, hipStream_t st);
int c1_wgrad_slabs(int N, const int cd[3]);
void igemm_parallelism(int dtype, const IgemmParams& p, long long* wgs, int* nkc);
int launch_ksplit_finish(int dtype, const IgemmParams& p, hipStream_t st);
void wgrad_choose_split(int dtype, WgradParams& p, int target = 0);   // target: workgroups wanted, 0 = default
int launch_pack_one(int dtype, const PackDesc& d, hipStream_t st);
int launch_norm_fwd(int dtype, const NormParams& p, hipStream_t st);
int launch_norm_bwd(int dtype, const NormParams& p, hipStream_t st);
bool finish_norm_ok(const IgemmParams& ip, int c, int npix);   // norm.hip: one launch can finish the split layer `ip` and normalise its output
int launch_finish_norm(int dtype, const IgemmParams& ip, const NormParams& p, hipStream_t st);
bool finish_norm_bwd_ok(const IgemmParams& ip, int dtype, int c, int npix, int nb);   // the same for a split data gradient in front of a norm backward
int launch_finish_norm_bwd(int dtype, const IgemmParams& ip, const NormParams& p, hipStream_t st);
int norm_bwd_reduce_blocks(int npix, int c);   // workgroups per image of the backward reduction
int launch_c1_s2_fwd(int dtype, const C1Params& p, hipStream_t st);
int launch_c1_s2_T(int dtype, const C1Params& p, hipStream_t st);
int launch_c1_s2_wgrad(int dtype, const C1Params& p, hipStream_t st);
bool c1_wgrad3d_mfma_applicable(int dtype, const C1Params& p);
int launch_stencil(const StencilParams& p, hipStream_t st);
bool stencil_norm_on_load_ok(const StencilParams& p);   // small.hip: the 2-D strip kernel serves p (it can normalise on load)
int launch_stencil_wgrad(const StencilParams& p, hipStream_t st);
void igemm_config(int dtype, const IgemmParams& p, int* cps, int* nfrag);
size_t igemm_lds_bytes(int dtype, const IgemmParams& p);
void wgrad_config(int dtype, const WgradParams& p, int* ff, int* sf, int* wf, int* wsn);
int wgrad_slabs(int dtype, const WgradParams& p);
size_t wgrad_lds_bytes(int dtype, const WgradParams& p);
}
#endif
