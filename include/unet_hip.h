/*
 * unet_hip.h -- C ABI of libunet_hip.so, the MI355X (gfx950) implementation of
 * the U-Net layer operators.
 *
 * This header is the drop-in boundary: plain pointers and sizes, no C++ or
 * torch types.  Each fp32 entry point replaces one host function template of
 * the reference's operator API (src/cuda_library/cuda_lib.h, cited per
 * function) and keeps that function's argument list, followed by one extra
 * argument: the HIP stream to enqueue on (NULL = the null stream).
 *
 * Conventions
 *   - every function returns UNET_OK (0) or an error code; the message of the
 *     last error on the calling thread is unet_last_error();
 *   - all pointers are device pointers unless the name says host;
 *   - calls only ENQUEUE work on `stream` (no device synchronisation) unless
 *     documented otherwise, so the host may run ahead of the GPU;
 *   - tensors are NCHW contiguous fp32 (index = ((n*C + c)*H + h)*W + w,
 *     cuda_lib.h:17-20); conv / transposed-conv weights are
 *     [C_in][C_out][kh][kw] followed by C_out biases in the same buffer; BN
 *     parameters are interleaved [gamma0,beta0,gamma1,beta1,...]; BN batch
 *     statistics are interleaved raw sums [sum x, sum x^2]; running statistics
 *     interleaved [mean,var]; Adam state interleaved [m,v] per parameter;
 *   - element counts are 64-bit, so tensors of 2^31 elements and more work
 *     (the reference's int counts do not).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_OK 0
#define UNET_ERR_INVALID 1     /* bad argument (shape, null pointer, ...) */
#define UNET_ERR_HIP 2         /* a HIP runtime call or kernel launch failed */
#define UNET_ERR_UNSUPPORTED 3 /* valid request this build has no kernel for */

typedef void *unet_stream_t; /* hipStream_t */
typedef void *unet_event_t;  /* hipEvent_t */

/* ---- runtime ----------------------------------------------------------- */

const char *unet_last_error(void);
int unet_version(void);

int unet_device_count(int *count);
int unet_set_device(int device);
int unet_get_device(int *device);
/* Replaces cudaCheckErr (cuda_ops.cu:566-578): hipGetLastError +
 * hipDeviceSynchronize, but returns the error instead of printing it. */
int unet_device_synchronize(void);
/* Replaces cudaGPUReset (cuda_ops.cu:554-564).  Synchronises the device and
 * drops the library's cached workspace; it does not call hipDeviceReset, so
 * other users of the device in the same process (e.g. torch) are unaffected. */
int unet_device_reset(void);

int unet_stream_create(unet_stream_t *stream);
int unet_stream_destroy(unet_stream_t stream);
int unet_stream_synchronize(unet_stream_t stream);
int unet_event_create(unet_event_t *event);
int unet_event_destroy(unet_event_t event);
int unet_event_record(unet_event_t event, unet_stream_t stream);
int unet_event_synchronize(unet_event_t event);
int unet_stream_wait_event(unet_stream_t stream, unet_event_t event);
int unet_event_elapsed_ms(float *ms, unet_event_t start, unet_event_t stop);

/* Per-kernel timing with HIP events recorded on the launch stream (no
 * counterpart in the reference).  unet_prof_enable(1) clears the records and
 * starts recording around the launches of the convolution / batch-norm /
 * pooling kernel families; unet_prof_get() synchronises on the recorded events
 * and returns, for family `kind` (0 <= kind < unet_prof_kinds), the number of
 * launches, their summed duration, and the algorithmic FLOPs and bytes the
 * launchers attributed to them. */
int unet_prof_enable(int on);
/* Same, but records only the families whose bit (1 << kind) is set in
 * kind_mask: timing one kernel family costs two event records per launch of
 * that family and nothing for the others. */
int unet_prof_enable_kinds(uint64_t kind_mask);
int unet_prof_kinds(int *count);
const char *unet_prof_name(int kind);
int unet_prof_get(int kind, int64_t *launches, double *total_ms, double *total_flops, double *total_bytes);

/* ---- memory ------------------------------------------------------------ */

/* Replaces cudaGetPointer<T>(n, ZERO) (cuda_ops.cu:32-57): allocate `bytes`
 * of device memory and zero it. */
int unet_malloc(void **ptr, size_t bytes);
/* Replaces cudaLibFree (cuda_ops.cu:508). */
int unet_free(void *ptr);
int unet_host_malloc(void **ptr, size_t bytes); /* pinned host memory */
int unet_host_free(void *ptr);
int unet_memset(void *ptr, int value, size_t bytes, unet_stream_t stream);
/* Replace cudaLibCopyToDevice / cudaLibCopyToHost / cudaLibCopyDeviceToDevice
 * (cuda_ops.cu:580-597, 677-682).  Asynchronous with respect to the host when
 * the host buffer is pinned; unet_memcpy_d2h is followed by a stream sync
 * when `blocking` is non-zero. */
int unet_memcpy_h2d(void *dst, const void *src_host, size_t bytes, unet_stream_t stream);
int unet_memcpy_d2h(void *dst_host, const void *src, size_t bytes, int blocking, unet_stream_t stream);
int unet_memcpy_d2d(void *dst, const void *src, size_t bytes, unet_stream_t stream);

/* Replaces cudaGetPointer<T>(n, HE, N, num_weights) (cuda_ops.cu:12-22,43-49):
 * weights[i] = normal(0,1) * sqrt(2/fan) for i < num_weights, from a
 * counter-based generator keyed by (seed, i).  The reference draws from
 * curand with seed 0, so values differ; the distribution is the same. */
int unet_init_he_f32(float *weights, int64_t num_weights, int64_t fan, uint64_t seed, unet_stream_t stream);
/* Replaces cudaGetPointer<T>(n, BN) (cuda_ops.cu:24-30,51-52):
 * w[i] = (i % 2 == 0), i.e. gamma = 1, beta = 0 interleaved. */
int unet_init_bn_f32(float *weights, int64_t num_weights, unet_stream_t stream);
/* Replaces cudaHasNans (cuda_ops.cu:649-675) with an on-device scan:
 * *has_nan_host = 1 when any element is NaN.  Synchronises `stream`. */
int unet_has_nans_f32(const float *ptr, int64_t num_floats, int *has_nan_host, unet_stream_t stream);
/* The same question without a synchronisation, for a check on every training step (the reference calls
 * cudaHasNans from inside its convolution layer when debugging, convolution_layer_functions.h:69-71):
 * *flag_device, an int in device memory that the caller zeroed once, becomes `value` (non-zero, e.g. the step
 * number) if it still was 0 and some element of ptr[0..num_floats) is NaN or +-Inf.  Asynchronous on `stream`;
 * read the flag whenever the host synchronises anyway.  ptr must be 16-byte aligned. */
int unet_flag_nonfinite_f32(const float *ptr, int64_t num_floats, int *flag_device, int value, unet_stream_t stream);

/* ---- convolution (cuda_lib.h:44-58, 162-177) ---------------------------- */

/* Replaces cudaConvolve.  Y[n,co,h,w] = b[co] + sum W[ci,co,kh,kw] *
 * X[n,ci,h*s-p+kh,w*s-p+kw], zero padding, square odd kernel. */
int unet_convolve_f32(const float *weights, const float *inputs, float *activations,
                      int64_t activation_array_length, int batch_size, int kernel_size,
                      int channels_in, int channels_out, int H_out, int W_out, int H_in,
                      int W_in, int padding, int stride, unet_stream_t stream);
/* Replaces cudaConvolveBackward.  grad_weights (weights then biases) is
 * overwritten.  grad_inputs may be NULL to skip the data gradient (first
 * layer). */
int unet_convolve_backward_f32(const float *grad_activations, const float *inputs,
                               const float *weights, float *grad_inputs, float *grad_weights,
                               int batch_size, int kernel_size, int channels_in,
                               int channels_out, int H_out, int W_out, int H_in, int W_in,
                               int padding, int stride, unet_stream_t stream);

/* ---- batch norm (cuda_lib.h:100-118, 179-189) --------------------------- */

/* Replaces cudaBatchNormTrain: raw sums -> BN_batch_stats; y = gamma*(x-mean)/
 * (sqrt(var)+1e-8)+beta with the biased batch variance; running [mean,var] =
 * 0.9*old + 0.1*batch.  BN_running_stats may be NULL. */
int unet_batch_norm_train_f32(const float *inputs, float *activations, const float *weights,
                              float *BN_batch_stats, float *BN_running_stats, int num_channels,
                              int H, int W, int batch_size, unet_stream_t stream);
/* cudaBatchNormTrain followed by cudaRelu (convolution_layer_functions.h:42-65) in one pass over the tensor:
 * activations = max(0, batch norm), the same values bit for bit; the pre-activation tensor is not written. */
int unet_batch_norm_train_relu_f32(const float *inputs, float *activations, const float *weights,
                                   float *BN_batch_stats, float *BN_running_stats, int num_channels,
                                   int H, int W, int batch_size, unet_stream_t stream);
/* unet_batch_norm_train_f32 for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments and results; the per-slice partial sums go to a fixed 1 MiB array inside the library's device
 * image instead of the grow-only workspace, and the call performs no hipMalloc / hipFree.  Shapes whose
 * partial sums do not fit (more than unet_batch_norm_train_noalloc_max_partials() (channel, slice) pairs;
 * at most 512 slices per channel are ever used, so any C <= 256 fits) return UNET_ERR_INVALID.  All calls on
 * one GPU share that GPU's array: calls that overlap on one GPU (different streams or threads) race.
 * The first call on each device looks the array's address up (hipGetSymbolAddress, which may load the code
 * object): make that call outside stream capture; every later call on the device only launches kernels.
 * With statistics synchronised across ranks (unet_bn_set_sync with a communicator) the entry point returns
 * UNET_ERR_INVALID: the cross-rank sum needs a staging buffer this variant does not have. */
int unet_batch_norm_train_noalloc_f32(const float *inputs, float *activations, const float *weights,
                                      float *BN_batch_stats, float *BN_running_stats, int num_channels,
                                      int H, int W, int batch_size, unet_stream_t stream);
int unet_batch_norm_train_noalloc_max_partials(void);
/* Replaces cudaBatchNormInference: y = gamma*(x-mean)/sqrt(var+1e-8)+beta with
 * the running statistics.  Extra argument batch_size (the reference handles
 * one image only). */
int unet_batch_norm_inference_f32(const float *inputs, float *activations, const float *weights,
                                  const float *BN_stats, int H, int W, int num_channels,
                                  int batch_size, unet_stream_t stream);
/* unet_batch_norm_inference_f32 followed by unet_relu_f32 in one pass over the tensor (the reference runs
 * cudaBatchNormInference and cudaRelu one after the other, convolution_layer_functions.h:59-71): the same
 * values bit for bit, one read and one write less per element. */
int unet_batch_norm_inference_relu_f32(const float *inputs, float *activations, const float *weights,
                                       const float *BN_stats, int H, int W, int num_channels,
                                       int batch_size, unet_stream_t stream);
/* Replaces cudaBatchNormBackward: grad_weights[2c] = sum dy*xhat,
 * grad_weights[2c+1] = sum dy; dx = gamma/sqrt(var+eps)*(dy - (dbeta +
 * xhat*dgamma)/M). */
int unet_batch_norm_backward_f32(const float *grad_activations, const float *inputs,
                                 const float *BN_batch_stats, const float *weights,
                                 float *grad_inputs, float *grad_weights, int num_channels,
                                 int H, int W, int batch_size, unet_stream_t stream);
/* cudaReluBackward followed by cudaBatchNormBackward (convolution_layer_functions.h:92-113) without the
 * intermediate tensor: grad_activations is the gradient behind the ReLU, activations that ReLU's OUTPUT (its
 * sign is the pre-activation's); same results bit for bit as the two calls. */
int unet_batch_norm_relu_backward_f32(const float *grad_activations, const float *activations, const float *inputs,
                                      const float *BN_batch_stats, const float *weights, float *grad_inputs,
                                      float *grad_weights, int num_channels, int H, int W, int batch_size,
                                      unet_stream_t stream);

/* ---- activations (cuda_lib.h:90-98, 151-161) ---------------------------- */

int unet_relu_f32(const float *inputs, float *activations, int64_t num_activations, unet_stream_t stream);
/* inputs = the PRE-activation values. */
int unet_relu_backward_f32(const float *grad_activations, const float *inputs, float *grad_inputs,
                           int64_t num_activations, unet_stream_t stream);
int unet_sigmoid_f32(const float *inputs, float *activations, int64_t num_activations, unet_stream_t stream);
/* activations = the sigmoid OUTPUT. */
int unet_sigmoid_backward_f32(const float *grad_activations, const float *activations,
                              float *grad_inputs, int64_t num_activations, unet_stream_t stream);
/* LEAKY_ReLU and TANH of enum Activation_type (layer.h:31), which the reference declares (LayerDesc carries
 * leaky_const, layer.h:36-44) but runs as ReLU (convolution_layer_functions.h:59-65): y = x > 0 ? x : slope*x
 * with the backward on the PRE-activation values like cudaReluBackward; y = tanh(x) with the backward on the
 * OUTPUT like cudaSigmoidBackward, dx = dy*(1 - y^2). */
int unet_leaky_relu_f32(const float *inputs, float *activations, int64_t num_activations, float slope,
                        unet_stream_t stream);
int unet_leaky_relu_backward_f32(const float *grad_activations, const float *inputs, float *grad_inputs,
                                 int64_t num_activations, float slope, unet_stream_t stream);
int unet_tanh_f32(const float *inputs, float *activations, int64_t num_activations, unet_stream_t stream);
int unet_tanh_backward_f32(const float *grad_activations, const float *activations, float *grad_inputs,
                           int64_t num_activations, unet_stream_t stream);

/* ---- max pool (cuda_lib.h:69-77, 202-211) -------------------------------- */

/* Replaces cudaMaxPool: k x k window at (ho*s, wo*s), no padding,
 * H_out = H/s, W_out = W/s. */
int unet_max_pool_f32(const float *inputs, float *activations, int batch_size, int stride,
                      int kernel_size, int in_channels, int in_height, int in_width,
                      unet_stream_t stream);
/* As unet_max_pool_f32, and also writes (either may be NULL)
 *   indices[n,c,ho,wo]: int32 flat index hi*W+wi of the first maximum in
 *     row-major window order (PyTorch's return_indices convention);
 *   tiemask[n,c,ho,wo]: bit (i*k+j) set when window element (i,j) equals the
 *     maximum (first 8 window positions). */
int unet_max_pool_indices_f32(const float *inputs, float *activations, int32_t *indices,
                              uint8_t *tiemask, int batch_size, int stride, int kernel_size,
                              int in_channels, int in_height, int in_width, unet_stream_t stream);
/* Replaces cudaMaxPoolBackward: dX = (X == Y[h/s,w/s]) ? dY[h/s,w/s] : 0, so
 * every tied maximum receives the gradient (max_pool.cu:172). */
int unet_max_pool_backward_f32(const float *grad_activations, const float *inputs,
                               const float *activations, float *grad_inputs, int batch_size,
                               int stride, int in_channels, int in_height, int in_width,
                               unet_stream_t stream);
/* Index-based backward (PyTorch semantics: only the saved argmax receives
 * the gradient).  No counterpart in the reference.  kernel_size <= stride. */
int unet_max_pool_backward_indices_f32(const float *grad_activations, const int32_t *indices,
                                       float *grad_inputs, int batch_size, int in_channels,
                                       int in_height, int in_width, int out_height,
                                       int out_width, unet_stream_t stream);

/* ---- attention gate ------------------------------------------------------ */
/* cudaAttention / cudaAttentionBackward (cuda_lib.h:121-132, 227-243;
 * attention.cu:9-499), same argument lists.  x [N,Cx,H_x,W_x] is gated by
 * g [N,Cg,H_x/2,W_x/2]:
 *   int_1 [N,Ci,H_x/2,W_x/2] = relu(b1 + 1x1 stride-2 projection of x + 1x1 projection of g)
 *   int_2 [N,H_x/2,W_x/2]    = sigmoid(bpsi + psi . int_1)
 *   activations [N,Cx,H_x,W_x] = x * blend(int_2), blend = the reference's 4x4-block mix of
 *   2x2 cells (attention.cu:61-78) with the "+1" cells clamped to the last row / column.
 * weights (layer.h:558-566): W[(Cx+Cg)][Ci] | psi[Ci] | b1[Ci] | bpsi =
 * unet_attention_param_count() floats.  All outputs are overwritten (the reference accumulates
 * into zeroed buffers with atomicAdd); every sum has a fixed order.  In the backward,
 * grad_activations_int_1 receives the gradient at the ReLU input and grad_activations_int_2
 * the gradient of int_2; grad_input_x includes the stride-2 projection's term. */
int unet_attention_param_count(int channels_in_x, int channels_in_g, int int_channels);
int unet_attention_f32(float *activations_int_1, float *activations_int_2, float *activations,
                       const float *weights, const float *input_x, const float *input_g,
                       int batch_size, int H_x, int W_x, int channels_in_x, int channels_in_g,
                       int int_channels, unet_stream_t stream);
int unet_attention_backward_f32(const float *grad_activations, const float *activations_int_1,
                                const float *activations_int_2, const float *input_x,
                                const float *input_g, const float *weights,
                                float *grad_activations_int_1, float *grad_activations_int_2,
                                float *grad_input_x, float *grad_input_g, float *grad_weights,
                                int batch_size, int H_x, int W_x, int channels_in_x,
                                int channels_in_g, int int_channels, unet_stream_t stream);

/* The same gate between NHWC bf16 tensors (pixel strides x_ld, g_ld, out_ld in elements; a tensor may be
 * a channel slice of a wider buffer): the projections, psi and their gradients run in fp32 on x gathered
 * at the even pixels, the passes over the full-resolution tensors read and write bf16 directly.
 * int_1 / int_2 are fp32 as above and must be kept for the backward.  The incoming gradient is dy (+ dy2
 * when not NULL); dx and dg are written, or added to when the accumulate flag is set. */
int unet_attention_bf16(const uint16_t *x, int x_ld, const uint16_t *g, int g_ld, const float *weights,
                        float *activations_int_1, float *activations_int_2, uint16_t *out, int out_ld,
                        int batch_size, int H_x, int W_x, int channels_in_x, int channels_in_g,
                        int int_channels, unet_stream_t stream);
int unet_attention_backward_bf16(const uint16_t *dy, int dy_ld, const uint16_t *dy2, int dy2_ld,
                                 const uint16_t *x, int x_ld, const uint16_t *g, int g_ld,
                                 const float *weights, const float *activations_int_1,
                                 const float *activations_int_2, uint16_t *dx, int dx_ld, int dx_accumulate,
                                 uint16_t *dg, int dg_ld, int dg_accumulate, float *grad_weights,
                                 int batch_size, int H_x, int W_x, int channels_in_x, int channels_in_g,
                                 int int_channels, unet_stream_t stream);

/* ---- upsample = transposed conv, kernel = stride = scale ---------------- */
/* (cuda_lib.h:79-88, 213-224) */

int unet_upsample_f32(const float *inputs, const float *weights, float *activations, int scale,
                      int num_in_channels, int num_out_channels, int H_in, int W_in,
                      int batch_size, unet_stream_t stream);
/* grad_weights (weights then biases) is overwritten; grad_inputs may be NULL. */
int unet_upsample_backward_f32(const float *grad_activations, const float *inputs,
                               const float *weights, float *grad_inputs, float *grad_weights,
                               int scale, int channels_in, int channels_out, int H_in, int W_in,
                               int batch_size, unet_stream_t stream);

/* ---- concat (cuda_lib.h:60-67, 191-198) ---------------------------------- */

/* Replaces cudaConcat: out[n,0:C1] = A[n], out[n,C1:C1+C2] = B[n]. */
int unet_concat_f32(const float *activations_1, const float *activations_2, float *activations,
                    int batch_size, int64_t HxW, int C1, int C2, unet_stream_t stream);
/* Replaces cudaConcatBackward.  dLdX receives the two parents' gradients back
 * to back: dLdX[0 : N*C1*HW] = dA as [N,C1,H,W], then dB as [N,C2,H,W] -- the
 * layout the reference's Layer::backward slices (layer.h:150-165).  (The
 * reference copies dLdY unchanged, which equals this only for batch 1.) */
int unet_concat_backward_f32(const float *dLdY, float *dLdX, int H, int W, int C1, int C2,
                             int batch_size, unet_stream_t stream);

/* cudaConcat for a first tensor that is LARGER than the second (skip connections of U-Nets built from
 * unpadded convolutions, which the reference's CONV descriptor allows but its concat rejects, layer.h:523-533):
 * A [N,C1,Ha,Wa] is centre-cropped to [Hb,Wb] (offsets (Ha-Hb)/2, (Wa-Wb)/2), then out[n,0:C1] = crop(A[n]),
 * out[n,C1:C1+C2] = B[n].  Backward: dA (full size, zero outside the window) and dB. */
int unet_concat_crop_f32(const float *activations_1, const float *activations_2, float *activations, int batch_size,
                         int C1, int C2, int H1, int W1, int H2, int W2, unet_stream_t stream);
int unet_concat_crop_backward_f32(const float *dLdY, float *dA, float *dB, int batch_size, int C1, int C2, int H1,
                                  int W1, int H2, int W2, unet_stream_t stream);

/* ---- losses (cuda_lib.h:32-36, 134-149) ---------------------------------- */

/* The kernel half of cudaDiceScore: loss_values[3n..3n+2] = [sum m*p, sum m,
 * sum p] per sample (deterministic; the reference uses float atomics). */
int unet_dice_sums_f32(const float *mask, const float *prediction, int H, int W, int batch_size,
                       float *loss_values, unet_stream_t stream);
/* The forward value of the difference loss, which the reference never computes (its train() prints the Dice
 * figures whatever the loss, neural_net.h:136-185): loss_values[3n..3n+2] = [sum (p-m)^2, sum m, sum p] per
 * sample; the loss is 0.5 * sum_n loss_values[3n] / (N*H*W), the function whose gradient
 * unet_difference_loss_backward_f32 returns. */
int unet_difference_loss_sums_f32(const float *mask, const float *prediction, int H, int W, int batch_size,
                                  float *loss_values, unet_stream_t stream);
/* Replaces cudaDiceScore: unet_dice_sums_f32, then copies the sums to the
 * host (synchronises `stream`) and returns mean_n 2*I/(sum m + sum p + 1e-5)
 * in *dice_host. */
int unet_dice_score_f32(const float *mask, const float *prediction, int H, int W, int batch_size,
                        float *loss_values, float *dice_host, unet_stream_t stream);
/* Replaces cudaDiceLossBackward: dLdY = scale * 2*(m*D - I)/D^2 with
 * D = sum m + sum p + 1e-5 read from loss_values.  scale = 1 reproduces the
 * reference; scale = -1/N gives the gradient of the loss 1 - mean dice. */
int unet_dice_loss_backward_f32(const float *mask, const float *prediction,
                                const float *loss_values, float *dLdY, int H, int W,
                                int batch_size, float scale, unet_stream_t stream);
/* Replaces cudaDifferenceLossBackward: dLdY = (y - mask)/total_size. */
int unet_difference_loss_backward_f32(const float *mask, const float *y, float *dLdY,
                                      int64_t total_size, unet_stream_t stream);
/* Softmax cross-entropy over the channel axis, mean over N*H*W pixels
 * (torch.nn.functional.cross_entropy semantics; the reference only names
 * SOFTMAX in an enum, layer.h:31).  logits [N,C,H,W], target int32 [N,H,W];
 * loss_out = device float (mean loss); dlogits may be NULL.  A target outside
 * [0, num_classes) marks its pixel as ignored (an "ignore label" such as 255 or
 * -1): no loss, zero gradient, and the mean is over the remaining pixels (0 if
 * there are none). */
int unet_softmax_xent_f32(const float *logits, const int32_t *target, float *dlogits,
                          float *loss_out, int batch_size, int num_classes, int H, int W,
                          unet_stream_t stream);
/* Helpers of the engine's cross-entropy loss (no counterpart in the reference): class indices handed over as
 * floats -> int32 (rounded to nearest; NaN and values beyond the int32 range become -1, which
 * unet_softmax_xent_f32 ignores); out[n][pixel] = arg max over the C channels of x[n][:][pixel] (first
 * on ties), as a float. */
int unet_f32_to_i32(const float *in, int32_t *out, int64_t n, unet_stream_t stream);
int unet_argmax_channels_f32(const float *x, float *out, int batch_size, int num_classes, int64_t HxW, unet_stream_t stream);

/* ---- optimiser and small ops (cuda_lib.h:29-30, 245-259) ----------------- */

/* Replaces cudaUpdateWeights (6 arguments): Adam, beta1 .9, beta2 .999,
 * eps 1e-8, decoupled weight decay 1e-4; t = 1-based step. */
int unet_update_weights_adam_f32(float *weights, const float *dLdW, float *adam_parameters,
                                 float learning_rate, int64_t num_weights, int t,
                                 unet_stream_t stream);
/* Replaces cudaUpdateWeights (4 arguments): w -= lr*g. */
int unet_update_weights_sgd_f32(float *weights, const float *dLdW, float learning_rate,
                                int64_t num_weights, unet_stream_t stream);
/* Replaces cudaAddBtoA: a += b. */
int unet_add_b_to_a_f32(float *a, const float *b, int64_t num_floats, unet_stream_t stream);
/* Replaces cudaThreshold: x = x >= t ? 1 : 0. */
int unet_threshold_f32(float *x, float threshold, int64_t num_elements, unet_stream_t stream);
/* Raw image input (no counterpart in the reference, whose caller normalises on the host, main.py:85-98):
 * uint8 [N][H][W][C] -> fp32 [N][C][H][W] with out = in*scale[c] + shift[c] (scale, shift in device memory);
 * uint8 -> fp32 with one scale (masks stored as 0/255). */
int unet_normalize_u8_nhwc_to_nchw_f32(const uint8_t *in, float *out, const float *scale, const float *shift, int N,
                                       int C, int H, int W, unet_stream_t stream);
int unet_u8_to_f32(const uint8_t *in, float *out, int64_t num_elements, float scale, unet_stream_t stream);
/* x *= alpha (gradient averaging across data-parallel ranks). */
int unet_scale_f32(float *x, float alpha, int64_t num_elements, unet_stream_t stream);

/* ---- data-parallel collectives (RCCL over xGMI) ---------------------------------
 * The reference trains on one device (no collective anywhere in src/); these entry points are what
 * batch-sharded training adds: one process per GPU, one communicator rank per process, the gradient mean
 * taken bucket by bucket on the communicator's own stream while backward continues.
 *   unet_comm_get_unique_id   rank 0 fills unet_comm_unique_id_bytes() bytes of host memory, which the caller
 *                             hands to every rank over any host channel (a file, MPI, torch.distributed, ...)
 *   unet_comm_init_rank       collective over all ranks; the communicator belongs to the current device
 *   unet_allreduce_begin      data[0:count] = mean over ranks, in place, enqueued on the communication stream
 *                             after `ready` (an event recorded on the compute stream once the bucket's last
 *                             gradient was written; NULL = no wait); returns at once
 *   unet_allreduce_wait       makes `stream` wait for every unet_allreduce_begin issued so far
 *   unet_allreduce_sum_f32, unet_broadcast_f32   in place, on `stream` itself
 *   unet_bn_set_sync          comm != NULL: the batch-norm training entry points (fp32 and bf16, forward and
 *                             backward) add their per-channel sums over the communicator's ranks, so that
 *                             every rank normalises with the statistics of the global batch and the result
 *                             equals single-GPU training on that batch (every rank must hold the same number
 *                             of elements; gamma/beta gradients come out divided by the number of ranks so
 *                             that the gradient mean is the global gradient).  NULL: per-rank statistics. */
typedef void *unet_comm_t;
int unet_comm_unique_id_bytes(void);
int unet_comm_get_unique_id(void *id_host);
int unet_comm_init_rank(unet_comm_t *comm, const void *id_host, int world_size, int rank);
int unet_comm_destroy(unet_comm_t comm);
int unet_comm_size(unet_comm_t comm, int *world_size, int *rank);
int unet_allreduce_begin(unet_comm_t comm, float *data, int64_t count, unet_event_t ready);
/* on != 0: unet_allreduce_begin sends its buckets as bf16 (rounded to nearest even on the communication stream,
 * averaged in bf16, widened back into `data`): half the bytes on the links and half the time the collective's
 * kernels share the GPU with backward, at bf16 precision of the averaged gradients.  Off by default. */
int unet_comm_set_bf16_gradients(unet_comm_t comm, int on);
/* Sizes the bf16 staging buffer for buckets of up to max_count elements, so that unet_allreduce_begin neither
 * allocates nor synchronises inside a backward pass (call it once after the buckets are known). */
int unet_comm_reserve_bf16_stage(unet_comm_t comm, int64_t max_count);
int unet_allreduce_wait(unet_comm_t comm, unet_stream_t stream);
int unet_allreduce_sum_f32(unet_comm_t comm, float *data, int64_t count, unet_stream_t stream);
int unet_broadcast_f32(unet_comm_t comm, float *data, int64_t count, int root, unet_stream_t stream);
int unet_bn_set_sync(unet_comm_t comm);

/* ==== bf16 / NHWC fast path =================================================
 * The training hot path of the network engine.  Activations are NHWC bf16
 * (uint16_t storage) with a pixel stride `ld` in elements, so a tensor may be
 * a channel slice of a wider buffer (the skip-connection concat is then free).
 * Channel counts are padded to multiples of 32 ("CiP", "CoP"); pad channels
 * hold zeros.  Accumulation is fp32 on the matrix cores.  These entry points
 * fuse what the reference runs as separate kernels (cudaConvolve +
 * cudaBatchNormTrain's statistics pass, cudaBatchNormTrain's normalise +
 * cudaRelu, cudaReluBackward + cudaBatchNormBackward, ...); their fp32
 * one-to-one counterparts are above. */

/* w_ref [Ci][Co][K][K] fp32 (reference layout) -> packed bf16 operands, zero padded, K*K*CoP*CiP elements each:
 *   K = 3 (convolution):  w_fwd [CiP/32][9][CoP][32]  (32-channel K chunks outermost),
 *                         w_bwd [CoP/32][9][CiP][32], taps flipped (data-gradient operand)
 *   K = 2 (the four phases of the 2x2 transposed convolution): w_fwd [4][CoP][CiP], w_bwd [4][CiP][CoP]
 * The layouts are what the convolution entry points below read; treat the buffers as opaque.
 * Either output may be NULL. */
int unet_pack_weights_bf16(const float *w_ref, uint16_t *w_fwd, uint16_t *w_bwd, int Ci, int Co, int CiP,
                           int CoP, int K, unet_stream_t stream);
/* The same for every layer of a network in one launch.  table_device (device memory) holds `layers` rows of ten
 * 64-bit integers {w_ref, w_fwd, w_bwd (addresses), Ci, Co, CiP, CoP, K, first_box, 0}: first_box = number of
 * 32x32 channel boxes, (CiP/32)*(CoP/32) per layer, of the rows before; total_boxes = their sum over all rows.
 * Per row the arguments mean what they mean for unet_pack_weights_bf16. */
int unet_pack_weights_batch_bf16(const int64_t *table_device, int layers, int total_boxes, unet_stream_t stream);
/* NCHW fp32 -> NHWC bf16 (pixel stride ld); channels [C, Cpad) are zeroed. */
int unet_nchw_f32_to_nhwc_bf16(const float *in, uint16_t *out, int N, int C, int H, int W, int ld, int Cpad,
                               unet_stream_t stream);
/* out[..., c] += in[n, c, ...] for c < C (the sum is rounded to bf16): adds an fp32 NCHW gradient
 * to an NHWC bf16 gradient buffer (cudaAddBtoA across the two layouts). */
int unet_nchw_f32_add_to_nhwc_bf16(const float *in, uint16_t *out, int N, int C, int H, int W, int ld,
                                   unet_stream_t stream);
int unet_nhwc_bf16_to_nchw_f32(const uint16_t *in, float *out, int N, int C, int H, int W, int ld,
                               unet_stream_t stream);

/* 3x3, padding 1, stride 1 convolution as an implicit GEMM on bf16 MFMA
 * (cudaConvolve, and with w_bwd / bias = NULL the data-gradient half of
 * cudaConvolveBackward).  out = conv(in) + bias, rounded to bf16; when
 * accumulate != 0 the result is added to what `out` holds (gradient fan-in; not together with stat_partials).
 * stat_partials (optional) receives rows of partial sums [rows][CoP][2] = (sum, sum of squares) of the
 * rounded outputs over disjoint sets of pixels, rows = unet_conv3x3_stat_rows(N, H, W, CiP, CoP) -- how the
 * pixels are grouped into rows depends on the kernel that serves the shape; consumers add the rows up
 * (unet_bn_finalize_bf16).  The count follows the kernel choice at the time of the call: it changes with
 * unet_conv3x3_pipeline_min_units, never exceeds max(N * unet_conv3x3_stat_tiles(H, W), 1024), and a caller that
 * allocates once sizes for that bound.  unet_conv3x3_stat_tiles(H, W) = number of 16x16 pixel tiles of one image (the
 * row count per image and phase of the transposed-convolution entry points). */
int unet_conv3x3_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                      uint16_t *out, int out_ld, float *stat_partials, int N, int H, int W, int CiP, int CoP,
                      int accumulate, unet_stream_t stream);
int unet_conv3x3_stat_rows(int N, int H, int W, int CiP, int CoP);
/* Two kernels serve the 3x3 entry points with bit-identical outputs: a persistent pipelined one (images of 32
 * rows and more, 64-channel blocks) and a general one.  The pipelined kernel is used when a call has at least
 * this many (32x16 or 16x16 pixel tile, 64-channel block) work units (default 128).  Sets the threshold when
 * min_units >= 0; returns the previous value.  Process-wide; meant for tests and tuning. */
int unet_conv3x3_pipeline_min_units(int min_units);
int unet_conv3x3_stat_tiles(int H, int W);
/* Inference: the same convolution with the batch norm (running statistics) and ReLU that follow it folded into the
 * epilogue (cudaConvolve + cudaBatchNormInference + cudaRelu in one kernel): out = relu(a*(conv + bias) + b) with
 * {a, b} from the per-channel records of unet_bn_inference_coef_bf16; the pre-BN tensor is never written. */
int unet_conv3x3_bn_relu_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                              const float *bn_coef, uint16_t *out, int out_ld, int N, int H, int W, int CiP, int CoP,
                              unet_stream_t stream);

/* The segmentation head: 3x3 / pad 1 / stride 1 convolution with ONE output channel (the last CONV_LAYER of
 * main.py:187-224; replaces cudaConvolve / cudaConvolveBackward, cuda_lib.h:44-256, for that layer).  The
 * one-channel tensor is fp32 [N][H][W] -- the layout of the fp32 BN / sigmoid / loss entry points -- so the head needs
 * no layout conversion and no channel padding.  CiP (padded input channels) must be 32, 64 or 128
 * (unet_conv3x3_head_supported); other widths use unet_conv3x3_bf16.
 *   unet_conv3x3_head_bf16        out[n,y,x] = bias[0] + sum in[n,y+kh-1,x+kw-1,ci] * W[ci][0][kh][kw];
 *                                 w_packed = the forward operand of unet_pack_weights_bf16 (K = 3)
 *   unet_conv3x3_head_dgrad_bf16  dx (NHWC bf16, CiP channels) from dz (fp32) and the fp32 weights w [Ci][1][3][3];
 *                                 with bn_z != NULL also the BN-backward pixel reduction of the layer that
 *                                 produced the head's input (arguments as unet_conv3x3_dgrad_bnred_bf16,
 *                                 bn_first_channel = 0); accumulate adds to dx (not with bn_z)
 *   unet_conv3x3_head_wgrad_bf16  grad_w [Ci][1][3][3] and grad_b[0] = sum dz (grad_b may be NULL), deterministic */
int unet_conv3x3_head_supported(int CiP);
int unet_conv3x3_head_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, float *out,
                           int N, int H, int W, int CiP, int CoP, unet_stream_t s);
int unet_conv3x3_head_dgrad_bf16(const float *dz, const float *w, int Ci, uint16_t *dx, int dx_ld, const uint16_t *bn_z,
                                 int bn_z_ld, const float *bn_coef, float *red_partials, int N, int H, int W, int CiP,
                                 int accumulate, unet_stream_t s);
int unet_conv3x3_head_wgrad_bf16(const float *dz, const uint16_t *x, int x_ld, float *grad_w, float *grad_b, int N,
                                 int H, int W, int CiP, int Ci, unet_stream_t s);
/* Data gradient of a 3x3 convolution whose INPUT is the output of a BN + ReLU
 * layer (fuses the first pass of cudaBatchNormBackward for that layer into the
 * data-gradient half of cudaConvolveBackward).  dx = conv(dz, w_packed_bwd) is
 * written (not accumulated), and red_partials receives
 * unet_conv3x3_stat_rows(N, H, W, CiP, CoP) rows [CoP][2] of partial sums
 * (sum g*xhat, sum g) with g = dx * relu'(a*z+b), where bn_z is that
 * layer's pre-BN tensor and bn_coef its unet_bn_finalize_bf16 records; feed
 * them to unet_bn_act_backward_from_partials_bf16.  When the input is a concat
 * and only its second half comes from such a layer, bn_first_channel is the
 * first channel of that half (a multiple of 64): bn_z, bn_coef and the
 * partials then cover channels [bn_first_channel, CoP) renumbered from 0, and
 * the partial rows are CoP - bn_first_channel channels wide. */
int unet_conv3x3_dgrad_bnred_bf16(const uint16_t *dz, int dz_ld, const uint16_t *w_packed_bwd, uint16_t *dx,
                                  int dx_ld, const uint16_t *bn_z, int bn_z_ld, const float *bn_coef,
                                  float *red_partials, int bn_first_channel, int N, int H, int W, int CiP,
                                  int CoP, unet_stream_t stream);
/* Weight-gradient half of cudaConvolveBackward: grad_w [Ci][Co][3][3] fp32
 * (overwritten) = sum over pixels of dz x shifted x.  Deterministic. */
int unet_conv3x3_wgrad_bf16(const uint16_t *dz, int dz_ld, const uint16_t *x, int x_ld, float *grad_w, int N, int H,
                            int W, int CiP, int CoP, int Ci, int Co, unet_stream_t stream);

/* 2x2 stride-2 transposed convolution (cudaUpsample / cudaUpsampleBackward):
 * out[n,2h+i,2w+j,co] = bias[co] + sum_ci in[n,h,w,ci] * W[ci,co,i,j].
 * stat_partials: [4][N][tiles][CoP][2] with tiles of the INPUT grid. */
int unet_tconv2x2_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                       uint16_t *out, int out_ld, float *stat_partials, int N, int H_in, int W_in, int CiP, int CoP,
                       unet_stream_t stream);
/* Inference: the transposed convolution with its batch norm (running statistics) and ReLU folded in, as
 * unet_conv3x3_bn_relu_bf16. */
int unet_tconv2x2_bn_relu_bf16(const uint16_t *in, int in_ld, const uint16_t *w_packed, const float *bias, int co_valid,
                               const float *bn_coef, uint16_t *out, int out_ld, int N, int H_in, int W_in, int CiP,
                               int CoP, unet_stream_t stream);
int unet_tconv2x2_dgrad_bf16(const uint16_t *dy, int dy_ld, const uint16_t *w_packed_bwd, uint16_t *dx, int dx_ld,
                             int N, int H_in, int W_in, int CiP, int CoP, int accumulate, unet_stream_t stream);
/* unet_tconv2x2_dgrad_bf16 (accumulate = 0) with the same fused BN-backward
 * reduction as unet_conv3x3_dgrad_bnred_bf16; red_partials is
 * [N][tiles(H_in, W_in)][CiP][2]. */
int unet_tconv2x2_dgrad_bnred_bf16(const uint16_t *dy, int dy_ld, const uint16_t *w_packed_bwd, uint16_t *dx,
                                   int dx_ld, const uint16_t *bn_z, int bn_z_ld, const float *bn_coef,
                                   float *red_partials, int N, int H_in, int W_in, int CiP, int CoP,
                                   unet_stream_t stream);
int unet_tconv2x2_wgrad_bf16(const uint16_t *dy, int dy_ld, const uint16_t *x, int x_ld, float *grad_w, int N,
                             int H_in, int W_in, int CiP, int CoP, int Ci, int Co, unet_stream_t stream);


/* Batch norm on NHWC bf16 tensors.  `coef` is a device array of
 * unet_bn_coef_floats(CoP) floats written by unet_bn_finalize_bf16 (training:
 * batch statistics from the convolution's per-tile partial sums; also writes
 * the raw sums to batch_stats[2C] and updates running_stats[2C] like
 * cudaBatchNormTrain) or unet_bn_inference_coef_bf16 (running statistics) and
 * read by the apply / backward kernels.  `count` = N*H*W. */
int unet_bn_coef_floats(int CoP);
int unet_bn_finalize_bf16(const float *stat_partials, int num_partials, int C, int CoP, int64_t count,
                          const float *gamma_beta, float *batch_stats, float *running_stats, float *coef,
                          unet_stream_t stream);
int unet_bn_inference_coef_bf16(const float *gamma_beta, const float *running_stats, int C, int CoP, float *coef,
                                unet_stream_t stream);
/* y = act(gamma*(z-mean)/(sqrt(var)+eps)+beta): the normalise half of
 * cudaBatchNormTrain / cudaBatchNormInference fused with cudaRelu (act 0),
 * cudaSigmoid (act 1) or nothing (act 2).  y may be a channel slice of a
 * concat buffer (y_ld > CoP). */
int unet_bn_act_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, int CoP,
                     int64_t pixels, int act, unet_stream_t stream);
/* The same with the activations the reference declares but runs as ReLU (layer.h:31): act 3 = LEAKY_ReLU with
 * slope act_param, act 4 = TANH (act 0..2 as above; act_param is ignored for them). */
int unet_bn_act_ex_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, int CoP,
                        int64_t pixels, int act, float act_param, unet_stream_t stream);
/* cudaReluBackward|cudaSigmoidBackward + cudaBatchNormBackward fused: the
 * incoming gradient is dy1 (+ dy2 when not NULL: gradient fan-in of a layer
 * with two consumers, replacing cudaAddBtoA); writes dz (bf16), grad_gb[2C]
 * = [dgamma,dbeta] interleaved and, when not NULL, grad_bias[C] = sum dz. */
int unet_bn_act_backward_bf16(const uint16_t *dy1, int ld1, const uint16_t *dy2, int ld2, const uint16_t *z, int z_ld,
                              const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld, float *grad_gb,
                              float *grad_bias, int C, int CoP, int64_t pixels, int act, unet_stream_t stream);
/* unet_bn_act_backward_bf16 for act 0..4 (3 = LEAKY_ReLU with slope act_param, 4 = TANH). */
int unet_bn_act_backward_ex_bf16(const uint16_t *dy1, int ld1, const uint16_t *dy2, int ld2, const uint16_t *z, int z_ld,
                                 const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld, float *grad_gb,
                                 float *grad_bias, int C, int CoP, int64_t pixels, int act, float act_param,
                                 unet_stream_t stream);

/* The same with the reduction pass already done by the kernel that produced
 * dy (unet_conv3x3_dgrad_bnred_bf16 / unet_tconv2x2_dgrad_bnred_bf16):
 * red_partials is [num_partials][CoP][2]; ReLU only. */
int unet_bn_act_backward_from_partials_bf16(const uint16_t *dy, int dy_ld, const uint16_t *z, int z_ld,
                                            const float *coef, const float *gamma_beta,
                                            const float *red_partials, int num_partials, uint16_t *dz,
                                            int dz_ld, float *grad_gb, float *grad_bias, int C, int CoP,
                                            int64_t pixels, unet_stream_t stream);

/* 2x2 stride-2 max pool on NHWC bf16 (cudaMaxPool) that also saves a tie mask:
 * mask[n,ho,wo,c] bit (2i+j) is set when x[n,2ho+i,2wo+j,c] equals the maximum.
 * The backward (cudaMaxPoolBackward) routes dy to every set bit. */
int unet_max_pool2x2_bf16(const uint16_t *x, int x_ld, uint16_t *y, int y_ld, uint8_t *mask, int N, int H, int W,
                          int C, unet_stream_t stream);
int unet_max_pool2x2_backward_bf16(const uint16_t *dy, int dy_ld, const uint8_t *mask, uint16_t *dx, int dx_ld, int N,
                                   int H, int W, int C, int accumulate, unet_stream_t stream);
/* unet_bn_act_bf16 followed by unet_max_pool2x2_bf16 in one pass over z (the
 * encoder's conv -> BN -> ReLU -> cudaMaxPool chain): writes the activation y,
 * the pooled activation and the tie mask; results identical to the two calls. */
int unet_bn_act_pool_bf16(const uint16_t *z, int z_ld, uint16_t *y, int y_ld, const float *coef, uint16_t *pooled,
                          int pooled_ld, uint8_t *mask, int N, int H, int W, int CoP, int act, unet_stream_t stream);
/* unet_max_pool2x2_backward_bf16 followed by unet_bn_act_backward_bf16 without
 * the full-resolution pool gradient in between (cudaMaxPoolBackward +
 * cudaAddBtoA + cudaReluBackward + cudaBatchNormBackward of a skip layer): the
 * incoming gradient is dy (may be NULL) plus pooled_dy routed through the tie
 * mask; z, dz and dy are [N][H][W] pixels, pooled_dy and mask [N][H/2][W/2].
 * Results identical to the two calls. */
int unet_bn_act_backward_pooled_bf16(const uint16_t *dy, int dy_ld, const uint16_t *pooled_dy, int pooled_ld,
                                     const uint8_t *mask, int N, int H, int W, const uint16_t *z, int z_ld,
                                     const float *coef, const float *gamma_beta, uint16_t *dz, int dz_ld,
                                     float *grad_gb, float *grad_bias, int C, int CoP, int act, unet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
