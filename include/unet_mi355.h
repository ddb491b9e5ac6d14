/* libunet_mi355 - C ABI of the MI355X (gfx950) U-Net hot path.
 *
 * The reference (hakkini2/UNet) has no FFI of its own: train.py reaches its arithmetic
 * through the Python surface of MONAI / PyTorch.  Every entry point below therefore cites
 * the reference call site whose work it takes over; the Python package `unet_amd` binds
 * these symbols with ctypes (unet_amd/_lib.py lists every prototype).
 *
 * Conventions
 *  - plain C types only; device buffers are raw pointers (torch `tensor.data_ptr()`),
 *    `stream` is a hipStream_t passed as void* (0 = default stream);
 *  - every function returns 0 on success, a negative code otherwise, and
 *    mi_last_error() then returns a thread-local message;
 *  - nothing allocates or frees device memory and nothing synchronises the stream:
 *    callers pass workspaces sized by the matching *_bytes() query;
 *  - activations are channels-last: element (n, z, y, x, ch) of a tensor lives at
 *    ((((n*D + z)*H + y)*W + x) * pitch + coff + ch); dtype 0 = fp32, 1 = bf16.
 *    One-channel tensors (image, logits, labels) are fp32 in PyTorch's contiguous layout.
 */
#ifndef UNET_MI355_H
#define UNET_MI355_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* mi_last_error(void);
int mi_version(void);

/* ------------------------------------------------------------------------------------------
 * Whole-network engine.
 * Replaces: monai.networks.nets.UNet construction (reference/train.py:78-86), model(inputs)
 * (train.py:115, 140) and the autograd backward of loss.backward() (train.py:117).
 * ------------------------------------------------------------------------------------------ */
typedef struct mi_unet_cfg {
    int32_t spatial_dims;      /* 2 | 3 */
    int32_t in_channels;       /* 1 */
    int32_t out_channels;      /* 1 */
    int32_t n_levels;          /* len(channels) >= 2 */
    int32_t channels[8];       /* multiples of 16 */
    int32_t strides[8];        /* all 2 */
    int32_t num_res_units;     /* >= 1 */
    float dropout;             /* probability, 0 <= p < 1 */
    int32_t dtype;             /* storage / MFMA input type of the hidden activations: 0 fp32, 1 bf16 */
    int32_t batch;             /* images per call */
    int32_t dims[3];           /* (D,H,W) of the input; D = 1 in 2-D; each divisible by 2^(n_levels-1) */
    int32_t with_backward;     /* 1: plan also holds the backward pass */
} mi_unet_cfg;

typedef struct mi_unet mi_unet;   /* opaque plan */

typedef struct mi_param_info {
    char name[160];            /* state_dict key, e.g. "model.0.conv.unit0.conv.weight" */
    int32_t ndim;
    int32_t shape[5];
    int64_t numel;
    int64_t offset;            /* element offset inside the flat fp32 parameter / gradient buffers */
    int32_t bucket;            /* gradient bucket in backward completion order: 0 decoder, 1 bottom, 2 encoder levels >= 1, 3 encoder level 0 */
} mi_param_info;

int mi_unet_create(const mi_unet_cfg* cfg, mi_unet** out);
void mi_unet_destroy(mi_unet* u);
int mi_unet_num_params(const mi_unet* u);                     /* parameter tensors (63 for the reference nets) */
int mi_unet_param_info(const mi_unet* u, int i, mi_param_info* info);   /* i in state_dict order */
int64_t mi_unet_param_floats(const mi_unet* u);               /* length of the flat buffers */
size_t mi_unet_workspace_bytes(const mi_unet* u);
#define MI_NUM_BUCKETS 4
int mi_unet_num_buckets(const mi_unet* u);                    /* MI_NUM_BUCKETS; a bucket may be empty (count 0) */
int mi_unet_bucket_range(const mi_unet* u, int bucket, int64_t* offset, int64_t* count);
double mi_unet_flops(const mi_unet* u, int backward);          /* 2*MAC of the convolutions of one call */
int mi_unet_num_steps(const mi_unet* u, int backward);         /* kernel launches per call (host logic check) */

/* logits[b,1,...] = UNet(x[b,1,...]); `training` != 0 enables dropout (seeded, stateless).  */
int mi_unet_forward(mi_unet* u, const float* params, const float* x, float* logits, void* workspace,
                    int training, uint64_t seed, void* stream);
/* grads (flat fp32, same layout as params) of the forward that last used `workspace`.
 * `x` is the forward's input (the first layer's weight gradient reads it);
 * `training` and `seed` must repeat the forward's values (the dropout mask is recomputed, not stored).
 * bucket < 0: whole backward; otherwise buckets must be run in order 0 .. MI_NUM_BUCKETS-1: when the call for
 * bucket i returns, everything that writes bucket i's range of `grads` has been enqueued on `stream`. */
int mi_unet_backward(mi_unet* u, const float* params, const float* x, const float* dlogits, float* grads,
                     void* workspace, int training, uint64_t seed, int bucket, void* stream);
/* The same for ONE bucket of a data-parallel step (reference/train.py:117 under a gradient exchange): the weight
 * gradients and the gradient reduction of the bucket run on the plan's internal stream, and when this call returns it
 * is `ready_stream` (the stream that will exchange the bucket), not `stream`, that waits for them: `stream` goes on with
 * the next bucket's chain at once instead of idling until the bucket's last weight gradient has finished (measured:
 * 2.53 -> 2.38 ms per step of the 32 x 512^2 benchmark in bucket-by-bucket mode, 2.37 ms as one call).  The caller makes `stream` wait for
 * `ready_stream` before anything on `stream` reads `grads` or reuses the workspace (the optimizer step, the next
 * forward).  ready_stream == NULL: exactly mi_unet_backward. */
int mi_unet_backward_bucket(mi_unet* u, const float* params, const float* x, const float* dlogits, float* grads,
                            void* workspace, int training, uint64_t seed, int bucket, void* stream, void* ready_stream);

/* ---- gradient exchange for hosts without torch.distributed ---------------------------------------------------------
 * No counterpart in the reference, which trains on one device (reference/train.py:75); this is the exchange step of the
 * data-parallel loop built on mi_unet_backward_bucket: one communicator per process and GPU, created with the device
 * current, RCCL underneath (resolved at run time; a process that already holds a copy of RCCL uses that one).  The
 * Python host reaches RCCL through torch.distributed instead (unet_amd/parallel.py) and does not call these.
 *   rank 0: mi_comm_unique_id(id) ; the host hands the 128 bytes to every rank (its own launcher / rendezvous)
 *   all   : mi_comm_init(nranks, rank, id, &c)                        collective: every rank must call it
 *   step  : for each bucket b, after mi_unet_backward_bucket(..., b, stream, comm_stream):
 *             mi_unet_bucket_range(u, b, &off, &cnt); mi_comm_allreduce_bucket(c, grads, off, cnt, comm_stream)
 *           then `stream` waits for `comm_stream`, and AdamW runs with grad_scale = 1 / nranks
 *   end   : mi_comm_destroy(c)
 * -3: RCCL is not available in this process (mi_last_error says what was tried). */
typedef struct mi_comm mi_comm;
int mi_comm_unique_id(void* id128);                                                  /* 128 bytes out */
int mi_comm_init(int nranks, int rank, const void* id128, mi_comm** out);
/* in-place sum over the ranks of grads[offset : offset + count] (fp32), enqueued on `stream`; count 0: nothing */
int mi_comm_allreduce_bucket(mi_comm* c, float* grads, int64_t offset, int64_t count, void* stream);
int mi_comm_destroy(mi_comm* c);

/* One optimisation step in one call: the body of the reference's inner loop, reference/train.py:111-118
 * (labels==2 -> 1 folded into the loss read, model(inputs), loss_function(outputs, labels), loss.backward(),
 * optimizer.step()), enqueued on `stream` from C: forward in training mode, Dice(+CE) loss forward and backward, the
 * whole backward pass and, when `do_adamw` is set, AdamW over the flat buffers.  Same kernels, same order and same
 * results, bit for bit, as mi_unet_forward + mi_dice_ce_loss_fwd + mi_dice_ce_loss_bwd + mi_unet_backward(bucket -1) +
 * mi_adamw_step with the same arguments; it exists because a step of a small batch is bound by the host when every
 * part is a call of its own from Python.  Data-parallel callers that exchange gradient buckets between the parts of
 * the backward pass keep using the separate entry points. */
typedef struct mi_train_step_args {
    float* params;             /* flat fp32 parameters (updated in place when do_adamw) */
    const float* x;            /* input batch [nb][1][...] fp32 */
    const void* target;        /* labels, same shape; dtype code below */
    float* logits;             /* [nb][1][...] fp32, written */
    float* dlogits;            /* same shape, scratch (gradient of the loss wrt the logits) */
    double* loss_sums;         /* [nb][4] scratch */
    float* loss;               /* [1], written */
    float* loss_accum;         /* optional running sum of the step losses (may be NULL) */
    float* grads;              /* flat fp32 gradients, written */
    float* exp_avg;            /* AdamW moments (flat fp32; unused unless do_adamw) */
    float* exp_avg_sq;
    void* workspace;           /* mi_unet_workspace_bytes */
    int32_t target_dtype;      /* 0 fp32, 1 uint8, 2 int64, 3 int32 */
    int32_t fold2;             /* label value 2 counts as 1 (train.py:112-113) */
    float smooth_nr, smooth_dr, lambda_dice, lambda_ce;
    int32_t do_adamw;
    float lr, beta1, beta2, eps, weight_decay, grad_scale;
    int64_t step;              /* AdamW step count of this update (>= 1) */
    uint64_t seed;             /* dropout seed of this step */
} mi_train_step_args;
int mi_unet_train_step(mi_unet* u, const mi_train_step_args* a, void* stream);

/* Per-launch timing: when enabled every kernel launch of the engine is bracketed by HIP events on
 * its stream.  mi_unet_prof_collect() synchronises those events and adds the elapsed times to the
 * per-step accumulators read by mi_unet_step_info(). */
int mi_unet_prof_enable(mi_unet* u, int on);
int mi_unet_prof_collect(mi_unet* u);
int mi_unet_prof_reset(mi_unet* u);
typedef struct mi_step_info {
    char layer[96];            /* which layer of the network the launch belongs to */
    char kernel[96];           /* kernel (template instance) it runs */
    double work;               /* algorithmic flops (convolutions) or bytes (everything else) per launch */
    int32_t work_is_flops;
    double bytes;              /* algorithmic HBM bytes per launch (tensors read once + written once) */
    double total_ms;           /* accumulated by mi_unet_prof_collect */
    int64_t launches;
} mi_step_info;
int mi_unet_step_info(const mi_unet* u, int backward, int i, mi_step_info* info);
/* debug aid: workspace offset of the tensor step i writes (-1: outside the workspace) */
int64_t mi_unet_step_output_offset(const mi_unet* u, int backward, int i);
int64_t mi_unet_step_aux_offset(const mi_unet* u, int backward, int i);   /* statistics block of a norm step */
/* Parity-test aid: where the k-th InstanceNorm + Dropout + PReLU block of the forward pass (in execution order =
 * the order in which MONAI's ADN modules run, reference/train.py:115) keeps its input in the workspace after
 * mi_unet_forward: the raw convolution output and the fp64 statistic slots [nb][8][c][2] {sum, sum of squares}.
 * From these a test recomputes which side of the PReLU kink every activation took. */
typedef struct mi_norm_info {
    int64_t raw_offset;        /* workspace byte offset of the raw conv output, channels-last [nb][npix][pitch] */
    int32_t pitch, coff, c;
    int32_t dtype;             /* element type of the raw tensor: 0 fp32, 1 bf16 */
    int64_t npix;              /* pixels per image at this level */
    int64_t stats_offset;      /* workspace byte offset of the statistic slots */
    uint32_t layer;            /* layer id that keys this block's dropout stream */
} mi_norm_info;
int mi_unet_num_norms(const mi_unet* u);
int mi_unet_norm_info(const mi_unet* u, int k, mi_norm_info* info);

/* ------------------------------------------------------------------------------------------
 * Loss / metric.  Replaces DiceCELoss(to_onehot_y=False, sigmoid=True)(outputs, labels) and its
 * backward (train.py:88,116-117,142), labels[labels==2]=1 (train.py:112-113,137-138), the
 * sigmoid + >=0.5 post transform and DiceMetric (train.py:89,92,145-152) and
 * calculate_dice_score (utils/utils.py:11-44).  Label dtype: 0 fp32, 1 uint8, 2 int64, 3 int32.
 * ------------------------------------------------------------------------------------------ */
int mi_dice_ce_loss_fwd(const float* logits, const void* target, int target_dtype, int fold2, int nb,
                        int64_t npix, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce,
                        double* sums /* [nb][4] scratch, kept for backward */, float* loss /* [1] */,
                        float* loss_accum /* optional running sum */, void* stream);
int mi_dice_ce_loss_bwd(const float* logits, const void* target, int target_dtype, int fold2, int nb,
                        int64_t npix, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce,
                        const double* sums, const float* grad_out /* optional device scalar */, float grad_scale,
                        float* dlogits, void* stream);
/* Loss forward for callers that cannot allocate.  No counterpart in the reference: it serves callers of this
 * library that may not call hipMalloc or bring a workspace (code under HIP graph capture, tight serving /
 * validation loops, harnesses that forbid allocation).  Same arithmetic and kernels as mi_dice_ce_loss_fwd, but the
 * per-sample sums live in a device array inside the library (128 KiB per GPU, held until the process ends), so
 * nb <= 4096 (larger batches are refused with an error) and no backward pass can follow from it.  Calls on one GPU
 * share that GPU's array: the array is cleared at the start of every call, and calls that overlap on one GPU race. */
int mi_dice_ce_loss_fwd_noalloc(const float* logits, const void* target, int target_dtype, int fold2, int nb,
                                int64_t npix, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce,
                                float* loss /* [1] */, float* loss_accum /* optional running sum */, void* stream);
int mi_dice_counts(const float* pred, int pred_is_binary, float thresh, const void* target, int target_dtype,
                   int fold2, int nb, int64_t npix, uint64_t* counts /* [nb][4] tp,fp,fn,tn */, void* stream);
int mi_fold_labels(void* target, int target_dtype, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer.  Replaces torch.optim.AdamW(...).step() (train.py:90,118) on the flat buffers.
 * ------------------------------------------------------------------------------------------ */
int mi_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Sliding-window inference helpers.  Replace the crop / blend arithmetic of
 * monai.inferers.sliding_window_inference (train.py:140); window placement is host logic.
 * `starts` is a device array [nwin][3] of window origins in the padded image.
 * ------------------------------------------------------------------------------------------ */
int mi_sw_gather(const float* image, float* windows, const int32_t* starts, int nwin, int win0, int nwin_launch,
                 int nb, const int32_t img[3], const int32_t roi[3], const int32_t pad_lo[3], float cval,
                 void* stream);
int mi_sw_blend(const float* win_out, float* out, const int32_t* starts, int nwin, int nb, const int32_t img[3],
                const int32_t roi[3], const int32_t pad_lo[3], void* stream);

/* ------------------------------------------------------------------------------------------
 * GPU-resident input pipeline of the 2-D training crops.  Replaces, on slices already in HBM, the
 * arithmetic of ScaleIntensityRanged, CropForegroundd, SpatialPadd, RandCropByPosNegLabeld and
 * RandRotate90d (reference/unetDataset.py:68-93); the random numbers are the caller's.
 * Slices are planar [S][H][W]; image dtype 0 fp32 | 1 int16; label dtype as above.
 * ------------------------------------------------------------------------------------------ */
/* bbox[s] = {y0, y1, x0, x1}, half open, of the pixels with img > thr (the whole slice when there is none) */
int mi_fg_bbox(const void* img, int img_dtype, float thr, int ns, int H, int W, int32_t* bbox, void* stream);
/* centers[b] = (y, x), relative to the bounding box of slice sample_slice[b], of the floor(u[b][1] * n)-th
 * pixel (row-major) of the class chosen by u[b][0] < pos_ratio: foreground = label > 0, background =
 * label <= 0 and img > bg_thr */
int mi_pick_centers(const void* img, int img_dtype, const void* lbl, int lbl_dtype, float bg_thr, const int32_t* bbox, int H,
                    int W, const int32_t* sample_slice, const float* u, float pos_ratio, int nsamp, int32_t* centers,
                    void* stream);
/* desc[b] = {slice, oy, ox, k}: crop origin in slice coordinates (outside the bounding box = zero padding)
 * and the number of counter-clockwise quarter turns; image values become
 * clip((v - a_min) / (a_max - a_min) * (b_max - b_min) + b_min) */
int mi_crop_scale_rot(const void* img, int img_dtype, const void* lbl, int lbl_dtype, int H, int W, const int32_t* desc,
                      const int32_t* bbox, int nsamp, float a_min, float a_max, float b_min, float b_max, int clip, int ch,
                      int cw, float* out_img, float* out_lbl, void* stream);

/* ------------------------------------------------------------------------------------------
 * Single operators (the engine is built from these; exported for operator-level parity tests
 * and for callers that assemble other networks).
 * ------------------------------------------------------------------------------------------ */
typedef struct mi_conv_desc {
    int32_t ndim;        /* 2 | 3 */
    int32_t transposed;  /* 0 conv (pad=(k-1)/2) | 1 transposed conv k3 s2 p1 output_padding 1 */
    int32_t ksize;       /* 3 | 1 */
    int32_t stride;      /* 1 | 2 */
    int32_t cin, cout;   /* multiples of 16 */
    int32_t nb;
    int32_t in_dims[3];  /* (D,H,W) of the layer input */
    int32_t dtype;       /* 0 fp32 | 1 bf16 */
} mi_conv_desc;

typedef struct mi_tensor {
    void* ptr; int32_t pitch; int32_t coff;
    void* ptr2; int32_t pitch2; int32_t coff2; int32_t c1;  /* optional: channels >= c1 come from ptr2 */
} mi_tensor;

/* torch weight layout in `w`: conv [cout][cin][k..], transposed conv [cin][cout][k..]; fp32 */
size_t mi_conv_packed_bytes(const mi_conv_desc* d);
int mi_conv_pack(const mi_conv_desc* d, int dgrad, const float* w, void* packed, void* stream);
/* F.conv{2,3}d / F.conv_transpose{2,3}d: y = conv(x) (+bias) (+addend); stats (optional) receives
 * per-(image, slot, channel) sum and sum of squares of the bias-free result ([nb][8][cout][2] fp64,
 * pre-zeroed; fp64 accumulators make the sums independent of workgroup arrival order) */
int mi_conv_forward(const mi_conv_desc* d, const mi_tensor* x, const void* packed, const float* bias,
                    const mi_tensor* y, const mi_tensor* addend, double* stats, void* stream);
int mi_conv_dgrad(const mi_conv_desc* d, const mi_tensor* dy, const void* packed_dgrad, const mi_tensor* dx,
                  const mi_tensor* addend, void* stream);
size_t mi_conv_wgrad_workspace_bytes(const mi_conv_desc* d);
/* dw in the torch layout, db [cout] (conv only; may be NULL) */
int mi_conv_wgrad(const mi_conv_desc* d, const mi_tensor* x, const mi_tensor* dy, float* dw, float* db,
                  void* workspace, void* stream);

/* Weight gradient of the 3^d stride-2 convolution between a ONE-channel fp32 tensor on the fine grid (x: [nb][fine_dims],
 * dense) and an n-channel channels-last tensor on the coarse grid (y, fine_dims / 2; ptr2 / c1 name a second source for
 * the channels from c1 on): dw[c][k] = sum_p y[p][c] * x[2p + k - 1], db[c] = sum_p y[p][c] (may be NULL).  This is the
 * first encoder convolution (x = image, y = gradient of its output; torch layout [n][1][k..]) and equally the last
 * transposed convolution (x = gradient of its one-channel output, y = its input; torch layout [n][1][k..] again).
 * The network engine runs these two layers on kernels of their own; this entry point reaches the same kernels. */
/* The same two layers forward.  mi_conv1_forward: y[p][c] = bias[c] + sum_k w[c][k] * x[2p + k - 1] (F.conv, k 3, stride 2,
 * padding 1, one input channel; w [n][1][k..] fp32, bias may be NULL; y->ptr2 / c1 name a second destination for the
 * channels from c1 on; stats as in mi_conv_forward, [nb][8][n][2]).  mi_conv1_transposed: x_out[2p + k - 1] += w[c][k] *
 * y[p][c], plus bias[0] (F.conv_transpose, k 3, stride 2, padding 1, output_padding 1, one output channel; w [n][1][k..];
 * stats [nb][8][1][2] of the bias-free result).  The data gradient of either layer is the other one's forward. */
int mi_conv1_forward(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const float* x, const float* w,
                     const float* bias, const mi_tensor* y, double* stats, void* stream);
int mi_conv1_transposed(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const mi_tensor* y, const float* w,
                        const float* bias, float* x_out, double* stats, void* stream);
size_t mi_conv1_wgrad_workspace_bytes(int ndim, int n, const int32_t* fine_dims);
int mi_conv1_wgrad(int dtype, int ndim, int nb, const int32_t* fine_dims, int n, const float* x, const mi_tensor* y,
                   float* dw, float* db, void* workspace, void* stream);

/* InstanceNorm(eps, no affine) -> Dropout(p) -> PReLU(alpha) (+ residual): MONAI ADN "NDA" */
int mi_instnorm_prelu_fwd(int dtype, int nb, int64_t npix, int c, const mi_tensor* raw, const double* stats,
                          const float* alpha, const mi_tensor* res, const mi_tensor* out, float dropout,
                          uint64_t seed, uint32_t layer, void* stream);
/* bstats: [nb][8][c][2] fp64 scratch (zeroed inside); dalpha: device scalar, accumulated */
int mi_instnorm_prelu_bwd(int dtype, int nb, int64_t npix, int c, const mi_tensor* dy, const mi_tensor* raw,
                          const double* stats, const float* alpha, double* bstats, float* dalpha,
                          const mi_tensor* dx, float dropout, uint64_t seed, uint32_t layer, void* stream);

#ifdef __cplusplus
}
#endif
#endif
