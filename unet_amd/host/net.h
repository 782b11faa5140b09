// Network engine: the layer graph behind `neural_net.NeuralNet`.
//
// Host-only C++ (no HIP headers): every device operation goes through the C ABI
// of libunet_hip.so (include/unet_hip.h).  It plays the role of the reference's
// class NN (src/neural_net_implementation/neural_net.h:23-265) and class Layer
// (src/neural_net_implementation/layer.h:46-617), redesigned:
//   - layers are kept in creation order, which the parent-index rule makes a
//     topological order, so forward is a loop and backward the reverse loop
//     (the reference recurses from the final layer);
//   - all parameters live in ONE flat buffer, all gradients in another, Adam
//     state in a third: one optimiser launch per step and contiguous buckets
//     for the data-parallel all-reduce;
//   - every buffer is allocated once per batch size; no malloc/free, no host
//     synchronisation and no device-to-host copy inside a training step.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <limits>
#include <vector>

#include "layer_desc.h"
#include "unet_hip.h"

namespace unet {

inline void check(int rc, const char *what) {
  if (rc != UNET_OK) {
    std::stringstream ss;
    ss << what << ": " << unet_last_error();
    throw std::runtime_error(ss.str());
  }
}
#define UNET_CALL(expr) ::unet::check((expr), #expr)

inline int64_t shape_size(const std::vector<int> &shape) {
  int64_t s = 1;
  for (int d : shape) s *= d;
  return s;
}

// Owning device allocation (zero-filled).
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : ptr_(o.ptr_), bytes_(o.bytes_) {
    o.ptr_ = nullptr;
    o.bytes_ = 0;
  }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
    if (this != &o) {
      release();
      ptr_ = o.ptr_;
      bytes_ = o.bytes_;
      o.ptr_ = nullptr;
      o.bytes_ = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  void ensure(size_t bytes) {
    if (bytes <= bytes_) return;
    release();
    UNET_CALL(unet_malloc(&ptr_, bytes));
    bytes_ = bytes;
  }
  void release() {
    if (ptr_) unet_free(ptr_);
    ptr_ = nullptr;
    bytes_ = 0;
  }
  float *f32() const { return static_cast<float *>(ptr_); }
  uint16_t *u16() const { return static_cast<uint16_t *>(ptr_); }
  uint8_t *u8() const { return static_cast<uint8_t *>(ptr_); }
  void *raw() const { return ptr_; }
  size_t bytes() const { return bytes_; }

 private:
  void *ptr_ = nullptr;
  size_t bytes_ = 0;
};

struct Node {
  Layer_type type;
  Activation_type activation = NONE;
  std::vector<int> parents;
  int num_children = 0;
  // geometry (per sample)
  int in_c = 0, in_h = 0, in_w = 0, out_c = 0, out_h = 0, out_w = 0;
  int kernel = 0, padding = 0, stride = 1;
  int c1 = 0, c2 = 0;  // concat: channels of the two parents
  int crop_h = 0, crop_w = 0;  // concat: size of the first parent when it is larger and gets centre-cropped (else 0)
  float leaky = 0.01f;         // LEAKY_ReLU slope (LayerDesc::leaky_const)
  std::vector<int> input_shape, output_shape;
  // parameters: slice [param_off, param_off + num_wb + num_bn) of the flat
  // buffers, laid out [weights | biases | gamma0,beta0,...] like the
  // reference's per-layer dLdW (layer.h:274)
  int64_t num_wb = 0, num_bn = 0, param_off = 0;
  int64_t fan = 0;
  DeviceBuffer running;      // [mean,var] interleaved
  DeviceBuffer batch_stats;  // raw sums
  // fp32 activations (NCHW): conv/tconv output, BN output, activation
  DeviceBuffer z, bn, act;
  DeviceBuffer grad_out;  // dL/d(act)
  bool grad_seen = false;
  // ---- bf16 / NHWC path ----
  int cip = 0, cop = 0;          // channel counts padded to multiples of 32
  bool narrow = false;           // final layer with C_out not a multiple of 32: BN/activation/loss run in fp32
  bool head = false;             // narrow 3x3 conv with one output channel: the dedicated head kernels apply
  DeviceBuffer zb;               // conv / tconv output before BN, [N,H,W,cop] bf16
  DeviceBuffer yb;               // activation, when it is not a slice of a concat buffer
  uint16_t *y_ptr = nullptr;     // where the activation lives ...
  int y_ld = 0;                  // ... and its pixel stride
  DeviceBuffer gb;               // gradient w.r.t. the activation from non-concat consumers, [N,H,W,ld] bf16
  bool g_written = false;
  const uint16_t *g2_ptr = nullptr;  // second gradient source: this layer's slice of a concat's gradient
  int g2_ld = 0;
  DeviceBuffer mask;             // max pool tie mask
  DeviceBuffer coef;             // batch-norm coefficients of the last training forward
  DeviceBuffer coef_inf;         // the same from the running statistics (inference), valid while Net::inf_coef_valid_
  DeviceBuffer w_fwd, w_bwd;     // packed bf16 weights
  DeviceBuffer stat_partials;
  DeviceBuffer bwd_partials;     // BN-backward sums emitted by the consumer's data-gradient kernel
  int bwd_partials_n = 0;        // number of partial rows in it; 0 = none this step
  int concat_child = -1;         // index of the CONCAT consumer (at most one on this path)
  int concat_slot = 0;           // 0: first parent (skip), 1: second parent
  int pool_child = -1;           // a MAX_POOL consumer whose forward runs inside this layer's BN + activation pass
  bool pooled_by_parent = false; // (pool) the parent's pass already wrote this layer's output and mask
  const Node *pool_src = nullptr;  // (backward) the pool whose gradient this layer still has to gather
  int64_t in_elems() const { return (int64_t)in_c * in_h * in_w; }
  int64_t out_elems() const { return (int64_t)out_c * out_h * out_w; }
  bool has_params() const { return type == CONV_LAYER || type == UPSAMPLING_LAYER; }  // weights + batch norm
  int att_ci() const { return in_c / 2; }  // attention gate: intermediate channels (attention_layer_functions.h:18)
};

class Net {
 public:
  Net() = default;
  ~Net() {
    for (void *e : bucket_events_) unet_event_destroy(e);
    for (InputSlot &sl : slots_) {
      if (sl.pinned) unet_host_free(sl.pinned);
      if (sl.copied) unet_event_destroy(sl.copied);
      if (sl.consumed) unet_event_destroy(sl.consumed);
    }
    if (copy_stream_) unet_stream_destroy(copy_stream_);
  }

  // ---- graph construction (neural_net.h:72-120, layer.h:412-616) ----------
  void create(const std::vector<LayerDesc> &desc, LossFunction loss) {
    if (!nodes_.empty()) throw std::runtime_error("create() was already called on this network");
    if (desc.empty()) throw std::runtime_error("your network does not have a single output node");
    loss_ = loss;
    const char *none = std::getenv("UNET_NONE_IS_IDENTITY");
    none_is_identity_ = none && none[0] && none[0] != '0';
    nodes_.reserve(desc.size());
    for (size_t i = 0; i < desc.size(); ++i) {
      for (int p : desc[i].parents) {
        if (p < 0 || p >= (int)i) {
          std::stringstream ss;
          ss << "invalid parent index: " << p << " is less than 0 or greater than current index " << i;
          throw std::runtime_error(ss.str());
        }
      }
      nodes_.emplace_back();
      try {
        init_node(nodes_.back(), desc[i]);
      } catch (...) {
        nodes_.clear();
        throw;
      }
    }
    std::set<int> is_parent;
    for (const LayerDesc &d : desc)
      for (int p : d.parents) is_parent.insert(p);
    if (is_parent.size() != desc.size() - 1) {
      nodes_.clear();
      throw std::runtime_error("your network does not have a single output node");
    }
    final_ = -1;
    for (size_t i = 0; i < desc.size(); ++i)
      if (!is_parent.count((int)i)) final_ = (int)i;
    for (const LayerDesc &d : desc)
      for (int p : d.parents) nodes_[p].num_children++;
    // input geometry = the (first) parentless layer
    for (const Node &n : nodes_)
      if (n.parents.empty()) {
        in_c_ = n.in_c;
        in_h_ = n.in_h;
        in_w_ = n.in_w;
        break;
      }
    for (const Node &n : nodes_)
      if (n.parents.empty() && (n.in_c != in_c_ || n.in_h != in_h_ || n.in_w != in_w_))
        throw std::runtime_error("all input layers must take the same [C,H,W] input");
    // parameters
    total_params_ = 0;
    for (Node &n : nodes_) {
      n.param_off = total_params_;
      total_params_ += n.num_wb + n.num_bn;
    }
    params_.ensure(sizeof(float) * (size_t)std::max<int64_t>(total_params_, 1));
    grads_.ensure(sizeof(float) * (size_t)std::max<int64_t>(total_params_, 1));
    UNET_CALL(unet_memset(grads_.raw(), 0, sizeof(float) * (size_t)std::max<int64_t>(total_params_, 1), stream_));
    adam_.ensure(sizeof(float) * 2 * (size_t)std::max<int64_t>(total_params_, 1));
    init_parameters(seed_);
    if (verbose_)
      for (const Node &n : nodes_)
        if (n.num_wb + n.num_bn > 0) std::cout << "add params: " << n.num_wb + n.num_bn << std::endl;
    // compute path: bf16 NHWC on the matrix cores when the graph allows it
    const char *want = std::getenv("UNET_DTYPE");
    const bool want_bf16 = !want || std::string(want) != "fp32";
    std::string why;
    use_bf16_ = want_bf16 && plan_bf16(why);
    if (want_bf16 && !use_bf16_ && verbose_)
      std::cout << "note: using the fp32 kernels (" << why << ")" << std::endl;
  }
  bool is_bf16() const { return use_bf16_; }

  // He(fan-in) weights, zero biases, gamma = 1, beta = 0, running mean 0 /
  // var 1.  (The reference's BN initialiser never runs, leaving gamma = 0:
  // layer.h:445-447 with cuda_ops.cu:41,52.)
  void init_parameters(uint64_t seed) {
    seed_ = seed;
    UNET_CALL(unet_memset(params_.raw(), 0, params_.bytes(), stream_));
    UNET_CALL(unet_memset(adam_.raw(), 0, adam_.bytes(), stream_));
    step_count_ = 0;
    uint64_t layer_id = 0;
    for (Node &n : nodes_) {
      ++layer_id;
      if (n.type == ATTENTION_LAYER) {
        // projection and psi weights He-initialised, the Ci + 1 biases zero (layer.h:562-569 passes the weight
        // count as the fan; the fan-in of the additive projection is used here)
        UNET_CALL(unet_init_he_f32(params_.f32() + n.param_off, n.num_wb - (n.att_ci() + 1), n.fan,
                                   seed * 1000003ull + layer_id, stream_));
        continue;
      }
      if (!n.has_params() && n.type != BN_LAYER && n.type != FC_LAYER) continue;
      float *w = params_.f32() + n.param_off;
      if (n.num_wb > 0) UNET_CALL(unet_init_he_f32(w, n.num_wb - n.out_c, n.fan, seed * 1000003ull + layer_id, stream_));
      if (n.num_bn == 0) continue;  // (FC layers have no batch norm)
      UNET_CALL(unet_init_bn_f32(w + n.num_wb, n.num_bn, stream_));
      n.running.ensure(sizeof(float) * (size_t)n.num_bn);
      n.batch_stats.ensure(sizeof(float) * (size_t)n.num_bn);
      // running [mean,var] = [0,1]: var sits at the odd slots
      UNET_CALL(unet_init_bn_f32(n.running.f32() + 1, n.num_bn - 1, stream_));
      UNET_CALL(unet_memset(n.running.raw(), 0, sizeof(float), stream_));
    }
    params_dirty_ = true;
    inf_coef_valid_ = false;
    pack_boxes_ = 0;
  }

  // ---- training step pieces (device pointers, nothing synchronises) --------
  // images [batch, C, H, W] fp32, masks [batch, H_out, W_out] fp32, on device.
  void forward_device(const float *images, int batch, bool inference) {
    if (batch < 1) throw std::runtime_error("Batch size must be greater than 0");
    require_created();
    ensure_batch(batch);
    if (use_bf16_) {
      if (params_dirty_) pack_all_weights();
      if (!inference) {
        inf_coef_valid_ = false;  // a training forward moves the running statistics
      } else if (!inf_coef_valid_) {
        // inference coefficients depend on gamma/beta and the running statistics only: computed when those
        // changed, not per forward (22 launches of a 52-launch batch-1 forward otherwise)
        for (Node &n : nodes_)
          if (n.has_params() && !n.narrow)
            UNET_CALL(unet_bn_inference_coef_bf16(params_.f32() + n.param_off + n.num_wb, n.running.f32(), n.out_c, n.cop,
                                                  n.coef_inf.f32(), stream_));
        inf_coef_valid_ = true;
      }
      UNET_CALL(unet_nchw_f32_to_nhwc_bf16(images, x16_.u16(), batch, in_c_, in_h_, in_w_, in_cp(), in_cp(), stream_));
      for (Node &n : nodes_) forward_node_bf16(n, batch, inference);
      return;
    }
    for (Node &n : nodes_) forward_node(n, images, batch, inference);
  }

  // Sums for the loss of the current forward output; fills loss_values_.
  void loss_forward_device(const float *masks, int batch) {
    Node &f = nodes_[final_];
    if (loss_ == LossFunction::kDiceLoss) {
      UNET_CALL(unet_dice_sums_f32(masks, f.act.f32(), f.out_h, f.out_w, batch, loss_values_.f32(), stream_));
    } else if (loss_ == LossFunction::kDifferenceLoss) {
      UNET_CALL(unet_difference_loss_sums_f32(masks, f.act.f32(), f.out_h, f.out_w, batch, loss_values_.f32(), stream_));
    } else {
      // cross-entropy: masks hold class indices; the loss (mean over the batch's pixels) goes to
      // loss_values_[0] and its gradient with respect to the logits straight into the final layer's grad_out
      const int64_t pixels = (int64_t)batch * f.out_h * f.out_w;
      labels_.ensure(sizeof(int32_t) * (size_t)pixels);
      UNET_CALL(unet_f32_to_i32(masks, static_cast<int32_t *>(labels_.raw()), pixels, stream_));
      UNET_CALL(unet_softmax_xent_f32(f.act.f32(), static_cast<const int32_t *>(labels_.raw()), f.grad_out.f32(),
                                      loss_values_.f32(), batch, f.out_c, f.out_h, f.out_w, stream_));
    }
  }

  void backward_device(const float *images, const float *masks, int batch) {
    require_created();
    Node &f = nodes_[final_];
    if (f.out_c != 1 && loss_ != LossFunction::kCrossEntropyLoss)
      throw std::runtime_error("the loss functions need a final layer with one output channel");
    for (Node &n : nodes_) n.grad_seen = false;
    next_bucket_ = 0;
    if (comm_) UNET_CALL(unet_allreduce_wait(comm_, stream_));  // a reduction of the last step still in flight
    // loss gradient -> final layer's grad_out
    if (loss_ == LossFunction::kDiceLoss) {
      // L = 1 - mean_n dice_n  =>  dL/dp = -(1/N) * 2(mD - I)/D^2
      UNET_CALL(unet_dice_loss_backward_f32(masks, f.act.f32(), loss_values_.f32(), f.grad_out.f32(), f.out_h,
                                            f.out_w, batch, -1.0f / (float)batch, stream_));
    } else if (loss_ == LossFunction::kDifferenceLoss) {
      UNET_CALL(unet_difference_loss_backward_f32(masks, f.act.f32(), f.grad_out.f32(),
                                                  (int64_t)batch * f.out_h * f.out_w, stream_));
    }  // cross-entropy: loss_forward_device has written the gradient
    f.grad_seen = true;
    if (use_bf16_)
      for (Node &n : nodes_) {
        n.g_written = false;
        n.g2_ptr = nullptr;
        n.pool_src = nullptr;
        n.bwd_partials_n = 0;
      }
    for (int i = (int)nodes_.size() - 1; i >= 0; --i) {
      if (use_bf16_)
        backward_node_bf16(nodes_[i], batch);
      else
        backward_node(nodes_[i], images, batch);
      record_finished_buckets(i);
    }
  }

  // One Adam step over every parameter (cuda_ops.cu:405-471 semantics, one
  // step counter for the whole network).
  void step_device(float learning_rate) {
    flag_nonfinite_gradients();  // (waits for every bucket's mean over the ranks first)
    ++step_count_;
    UNET_CALL(unet_update_weights_adam_f32(params_.f32(), grads_.f32(), adam_.f32(), learning_rate,
                                           total_params_, step_count_, stream_));
    params_dirty_ = true;
    inf_coef_valid_ = false;
  }

  // ---- per-step NaN / Inf watch (UNET_CHECK_NAN=1 or set_check_nan): one asynchronous pass over the gradient
  // vector per step records the first optimizer step whose gradients were not finite; the host looks at the
  // record where it synchronises anyway (last_loss, the end of an epoch) and raises there.
  void set_check_nan(bool on) { check_nan_ = on; }
  bool check_nan() const { return check_nan_; }
  void flag_nonfinite_gradients() {
    wait_gradients();
    if (!check_nan_) return;
    nan_flag_.ensure(sizeof(int));
    UNET_CALL(unet_flag_nonfinite_f32(grads_.f32(), total_params_, static_cast<int *>(nan_flag_.raw()), step_count_ + 1,
                                      stream_));
  }
  // 0, or the optimizer step (1-based) at which non-finite gradients were first seen.  Synchronises.
  int nonfinite_step() {
    if (!nan_flag_.raw()) return 0;
    int v = 0;
    UNET_CALL(unet_memcpy_d2h(&v, nan_flag_.raw(), sizeof(int), 1, stream_));
    return v;
  }
  void raise_if_nonfinite() {
    if (!check_nan_) return;
    const int at = nonfinite_step();
    if (at) throw std::runtime_error("NaN or Inf in the gradients, first seen at optimizer step " + std::to_string(at));
  }

  // ---- host-facing helpers ----------------------------------------------------
  // Copies a batch from host memory and runs forward (+ loss sums, backward,
  // step when training).  Returns nothing; results are read with the getters.
  void train_batch_host(const float *images, const float *masks, int batch, float lr, bool do_step) {
    ensure_batch(batch);
    const size_t ib = sizeof(float) * (size_t)batch * in_c_ * in_h_ * in_w_;
    const Node &f = nodes_[final_];
    const size_t mb = sizeof(float) * (size_t)batch * f.out_h * f.out_w;
    UNET_CALL(unet_memcpy_h2d(in_images_.raw(), images, ib, stream_));
    UNET_CALL(unet_memcpy_h2d(in_masks_.raw(), masks, mb, stream_));
    forward_device(in_images_.f32(), batch, false);
    loss_forward_device(in_masks_.f32(), batch);
    backward_device(in_images_.f32(), in_masks_.f32(), batch);
    if (do_step)
      step_device(lr);
    else
      flag_nonfinite_gradients();
  }

  // ---- double-buffered input feed for train(): while the GPU works on batch b, the host copies
  // batch b+1 into pinned memory and a second stream moves it to the device (the reference copies
  // each batch from pageable memory on the compute stream, neural_net.h:160-166).
  void stage_batch_host(int slot, const float *images, const float *masks, int batch) {
    ensure_batch(batch);
    InputSlot &sl = slots_[slot & 1];
    const size_t ib = sizeof(float) * (size_t)batch * in_c_ * in_h_ * in_w_;
    const Node &f = nodes_[final_];
    const size_t mb = sizeof(float) * (size_t)batch * f.out_h * f.out_w;
    if (!copy_stream_) UNET_CALL(unet_stream_create(&copy_stream_));
    if (!sl.copied) {
      UNET_CALL(unet_event_create(&sl.copied));
      UNET_CALL(unet_event_create(&sl.consumed));
    }
    if (sl.pinned_bytes < ib + mb) {
      if (sl.pinned) {
        UNET_CALL(unet_event_synchronize(sl.copied));
        UNET_CALL(unet_host_free(sl.pinned));
      }
      UNET_CALL(unet_host_malloc(&sl.pinned, ib + mb));
      sl.pinned_bytes = ib + mb;
    }
    sl.images.ensure(ib);
    sl.masks.ensure(mb);
    if (sl.in_use) UNET_CALL(unet_event_synchronize(sl.copied));  // the pinned buffer is free again
    std::memcpy(sl.pinned, images, ib);
    std::memcpy(static_cast<char *>(sl.pinned) + ib, masks, mb);
    if (sl.in_use) UNET_CALL(unet_stream_wait_event(copy_stream_, sl.consumed));  // device buffers no longer read
    UNET_CALL(unet_memcpy_h2d(sl.images.raw(), sl.pinned, ib, copy_stream_));
    UNET_CALL(unet_memcpy_h2d(sl.masks.raw(), static_cast<char *>(sl.pinned) + ib, mb, copy_stream_));
    UNET_CALL(unet_event_record(sl.copied, copy_stream_));
    sl.in_use = true;
    sl.batch = batch;
  }
  // Per-channel input normalisation of the byte feed: x -> (x/255 - mean[c]) / (std[c] + 1e-7)  (main.py:89-92)
  void set_input_normalization(const float *mean, const float *stddev, int channels) {
    require_created();
    if (channels != in_c_) throw std::runtime_error("set_input_normalization: one mean / std per input channel");
    std::vector<float> coef(2 * (size_t)channels);
    for (int c = 0; c < channels; ++c) {
      const float inv = 1.0f / (stddev[c] + 1e-7f);
      coef[c] = inv / 255.0f;
      coef[channels + c] = -mean[c] * inv;
    }
    norm_coef_.ensure(sizeof(float) * coef.size());
    UNET_CALL(unet_memcpy_h2d(norm_coef_.raw(), coef.data(), sizeof(float) * coef.size(), stream_));
    sync();  // `coef` is a local
    norm_set_ = true;
  }
  // The same double-buffered feed for raw images: uint8 [N][H][W][C] images and uint8 masks cross PCIe as
  // bytes (a quarter of the fp32 volume); the copy stream then normalises them into the slot's fp32 buffers.
  void stage_batch_host_u8(int slot, const uint8_t *images, const uint8_t *masks, int batch, float mask_scale) {
    ensure_batch(batch);
    if (!norm_set_) throw std::runtime_error("stage_batch_host_u8: call set_input_normalization first");
    InputSlot &sl = slots_[slot & 1];
    const Node &f = nodes_[final_];
    const size_t ie = (size_t)batch * in_c_ * in_h_ * in_w_, me = (size_t)batch * f.out_h * f.out_w;
    if (!copy_stream_) UNET_CALL(unet_stream_create(&copy_stream_));
    if (!sl.copied) {
      UNET_CALL(unet_event_create(&sl.copied));
      UNET_CALL(unet_event_create(&sl.consumed));
    }
    if (sl.pinned_bytes < ie + me) {
      if (sl.pinned) {
        UNET_CALL(unet_event_synchronize(sl.copied));
        UNET_CALL(unet_host_free(sl.pinned));
      }
      UNET_CALL(unet_host_malloc(&sl.pinned, ie + me));
      sl.pinned_bytes = ie + me;
    }
    sl.images.ensure(sizeof(float) * ie);
    sl.masks.ensure(sizeof(float) * me);
    sl.raw.ensure(ie + me);
    if (sl.in_use) UNET_CALL(unet_event_synchronize(sl.copied));  // the pinned buffer is free again
    std::memcpy(sl.pinned, images, ie);
    std::memcpy(static_cast<char *>(sl.pinned) + ie, masks, me);
    if (sl.in_use) UNET_CALL(unet_stream_wait_event(copy_stream_, sl.consumed));  // device buffers no longer read
    UNET_CALL(unet_memcpy_h2d(sl.raw.raw(), sl.pinned, ie + me, copy_stream_));
    const uint8_t *raw = static_cast<const uint8_t *>(sl.raw.raw());
    UNET_CALL(unet_normalize_u8_nhwc_to_nchw_f32(raw, sl.images.f32(), norm_coef_.f32(), norm_coef_.f32() + in_c_, batch,
                                                 in_c_, in_h_, in_w_, copy_stream_));
    UNET_CALL(unet_u8_to_f32(raw + ie, sl.masks.f32(), (int64_t)me, mask_scale, copy_stream_));
    UNET_CALL(unet_event_record(sl.copied, copy_stream_));
    sl.in_use = true;
    sl.batch = batch;
  }
  // Forward pass (inference) on a staged batch; the output is read with the usual getters.
  void infer_batch_staged(int slot) {
    InputSlot &sl = slots_[slot & 1];
    if (!sl.in_use) throw std::runtime_error("infer_batch_staged: the slot holds no batch");
    UNET_CALL(unet_stream_wait_event(stream_, sl.copied));
    forward_device(sl.images.f32(), sl.batch, true);
    UNET_CALL(unet_event_record(sl.consumed, stream_));
  }
  // One training step on a staged batch; returns the device pointer of its masks' loss sums owner.
  void train_batch_staged(int slot, float lr, bool do_step) {
    InputSlot &sl = slots_[slot & 1];
    if (!sl.in_use) throw std::runtime_error("train_batch_staged: the slot holds no batch");
    UNET_CALL(unet_stream_wait_event(stream_, sl.copied));
    forward_device(sl.images.f32(), sl.batch, false);
    loss_forward_device(sl.masks.f32(), sl.batch);
    backward_device(sl.images.f32(), sl.masks.f32(), sl.batch);
    if (do_step) step_device(lr);
    UNET_CALL(unet_event_record(sl.consumed, stream_));
  }

  // Loss of the last batch (synchronises): Dice: 1 - mean dice; difference:
  // 0.5 * mean (y - m)^2.
  float last_loss(int batch, float *dice_out = nullptr) {
    const Node &f = nodes_[final_];
    sync();
    std::vector<float> lv(3 * (size_t)batch);
    UNET_CALL(unet_memcpy_d2h(lv.data(), loss_values_.raw(), sizeof(float) * lv.size(), 1, stream_));
    raise_if_nonfinite();
    if (loss_ == LossFunction::kCrossEntropyLoss) {
      if (dice_out) *dice_out = std::numeric_limits<float>::quiet_NaN();
      return lv[0];
    }
    if (loss_ == LossFunction::kDiceLoss) {
      float dice = 0.f;
      for (int i = 0; i < batch; ++i) dice += 2.0f * lv[3 * i] / (lv[3 * i + 1] + lv[3 * i + 2] + 1e-5f);
      dice /= (float)batch;
      if (dice_out) *dice_out = dice;
      return 1.0f - dice;
    }
    double s = 0;
    for (int i = 0; i < batch; ++i) s += lv[3 * i];
    if (dice_out) *dice_out = std::numeric_limits<float>::quiet_NaN();  // not measured under this loss
    return (float)(0.5 * s / ((double)batch * f.out_h * f.out_w));
  }

  // With a communicator attached the gradient buckets are averaged on the communication stream; whoever reads
  // the gradient buffer on the compute stream (or after a host synchronisation) first makes that stream wait
  // for the reductions issued so far.
  void wait_gradients() {
    if (comm_) UNET_CALL(unet_allreduce_wait(comm_, stream_));
  }
  void sync() {
    wait_gradients();
    UNET_CALL(unet_stream_synchronize(stream_));
  }

  void copy_params_to_host(float *dst) {
    UNET_CALL(unet_memcpy_d2h(dst, params_.raw(), sizeof(float) * (size_t)total_params_, 1, stream_));
  }
  void copy_params_from_host(const float *src) {
    UNET_CALL(unet_memcpy_h2d(params_.raw(), src, sizeof(float) * (size_t)total_params_, stream_));
    sync();
    params_dirty_ = true;
    inf_coef_valid_ = false;
  }
  void copy_grads_to_host(float *dst) {
    wait_gradients();
    UNET_CALL(unet_memcpy_d2h(dst, grads_.raw(), sizeof(float) * (size_t)total_params_, 1, stream_));
  }
  void copy_output_to_host(int layer, float *dst, int batch) {
    Node &n = nodes_.at(layer);
    const size_t bytes = sizeof(float) * (size_t)batch * n.out_elems();
    if (use_bf16_ && !n.narrow) {
      // NHWC bf16 (possibly a slice of a concat buffer) -> NCHW fp32
      n.act.ensure(bytes);
      UNET_CALL(unet_nhwc_bf16_to_nchw_f32(n.y_ptr, n.act.f32(), batch, n.out_c, n.out_h, n.out_w, n.y_ld, stream_));
    }
    UNET_CALL(unet_memcpy_d2h(dst, n.act.raw(), bytes, 1, stream_));
  }
  void copy_running_stats_to_host(int layer, float *dst) {
    const Node &n = nodes_.at(layer);
    UNET_CALL(unet_memcpy_d2h(dst, n.running.raw(), sizeof(float) * (size_t)n.num_bn, 1, stream_));
  }

  void copy_running_stats_from_host(int layer, const float *src) {
    const Node &n = nodes_.at(layer);
    UNET_CALL(unet_memcpy_h2d(n.running.raw(), src, sizeof(float) * (size_t)n.num_bn, stream_));
    sync();
    inf_coef_valid_ = false;
  }
  // Adam state ([m,v] interleaved per parameter) and step count, for checkpoints.
  void copy_optimizer_state_to_host(float *dst) {
    UNET_CALL(unet_memcpy_d2h(dst, adam_.raw(), 2 * sizeof(float) * (size_t)total_params_, 1, stream_));
  }
  void copy_optimizer_state_from_host(const float *src, int step_count) {
    UNET_CALL(unet_memcpy_h2d(adam_.raw(), src, 2 * sizeof(float) * (size_t)total_params_, stream_));
    sync();
    step_count_ = step_count;
  }

  // ---- data-parallel support ---------------------------------------------------
  // Gradient buckets: contiguous slices of the flat gradient buffer, cut at
  // layer boundaries, numbered in the order backward completes them (last layer
  // first).  After backward_device() bucket b is final once its event fires.
  void set_bucket_bytes(int64_t bytes) {
    bucket_bytes_ = bytes;
    buckets_.clear();
    reserve_comm_stage();
  }
  struct Bucket {
    int64_t offset, count;  // in floats
    int first_layer;        // the bucket is final when this layer's backward is done
  };
  const std::vector<Bucket> &buckets() {
    if (buckets_.empty() && total_params_ > 0) {
      int64_t end = total_params_;
      int64_t cur = 0;
      for (int i = (int)nodes_.size() - 1; i >= 0; --i) {
        const Node &n = nodes_[i];
        cur += n.num_wb + n.num_bn;
        const bool last = (i == 0);
        if ((cur * (int64_t)sizeof(float) >= bucket_bytes_ || last) && cur > 0) {
          buckets_.push_back(Bucket{n.param_off, end - n.param_off, i});
          end = n.param_off;
          cur = 0;
        }
      }
      while (bucket_events_.size() < buckets_.size()) {
        unet_event_t e;
        UNET_CALL(unet_event_create(&e));
        bucket_events_.push_back(e);
      }
    }
    return buckets_;
  }
  void *bucket_event(size_t i) { return bucket_events_.at(i); }
  // Data-parallel training over a communicator of the C ABI (unet_comm_init_rank; NULL = single process):
  // backward_device() then starts the all-reduce of every gradient bucket as it completes and
  // step_device() waits for all of them.
  void set_comm(unet_comm_t comm) {
    comm_ = comm;
    reserve_comm_stage();
  }
  // room for the largest bucket in the communicator's bf16 staging buffer (used when the buckets travel as bf16),
  // so that no reduction started from backward has to allocate
  void reserve_comm_stage() {
    if (!comm_) return;
    int64_t largest = 0;
    for (const Bucket &b : buckets()) largest = std::max<int64_t>(largest, b.count);
    if (largest > 0) UNET_CALL(unet_comm_reserve_bf16_stage(comm_, largest));
  }
  unet_comm_t comm() const { return comm_; }
  // Rank `root`'s parameters, running statistics and optimiser state to every rank.
  void broadcast_state(int root) {
    if (!comm_) return;
    UNET_CALL(unet_broadcast_f32(comm_, params_.f32(), total_params_, root, stream_));
    UNET_CALL(unet_broadcast_f32(comm_, adam_.f32(), 2 * total_params_, root, stream_));
    for (Node &n : nodes_)
      if (n.num_bn > 0) UNET_CALL(unet_broadcast_f32(comm_, n.running.f32(), n.num_bn, root, stream_));
    parameters_changed();
  }
  // To be called after the parameter buffer (param_ptr()) was written from outside the engine: the packed
  // bf16 weights and the cached inference batch-norm coefficients are rebuilt before their next use.
  void parameters_changed() {
    params_dirty_ = true;
    inf_coef_valid_ = false;
  }

  // ---- accessors -----------------------------------------------------------------
  bool created() const { return !nodes_.empty(); }
  const std::vector<Node> &nodes() const { return nodes_; }
  int final_layer() const { return final_; }
  int64_t num_parameters() const { return total_params_; }
  float *param_ptr() const { return params_.f32(); }
  float *grad_ptr() const { return grads_.f32(); }
  float *loss_values_ptr() const { return loss_values_.f32(); }
  int in_c() const { return in_c_; }
  int in_h() const { return in_h_; }
  int in_w() const { return in_w_; }
  unet_stream_t stream() const { return stream_; }
  void set_stream(unet_stream_t s) { stream_ = s; }
  void set_verbose(bool v) { verbose_ = v; }
  bool verbose() const { return verbose_; }
  LossFunction loss() const { return loss_; }
  int step_count() const { return step_count_; }
  void set_seed(uint64_t s) { seed_ = s; }

  void ensure_batch(int batch) {
    require_created();
    if (batch <= alloc_batch_) return;
    const size_t b = (size_t)batch;
    int64_t max_out = 0, max_in = 0;
    if (use_bf16_) {
      ensure_batch_bf16(batch);
      alloc_batch_ = batch;
      return;
    }
    for (Node &n : nodes_) {
      const size_t out_bytes = sizeof(float) * b * (size_t)n.out_elems();
      if (n.has_params()) {
        n.z.ensure(out_bytes);
        n.bn.ensure(out_bytes);
      }
      if (n.type == BN_LAYER || n.type == FC_LAYER) n.bn.ensure(out_bytes);  // the activation's argument
      n.act.ensure(out_bytes);
      n.grad_out.ensure(out_bytes);
      max_out = std::max(max_out, n.out_elems());
      int64_t in_e = n.type == CONCAT_LAYER ? n.out_elems() : n.in_elems();
      if (n.type == CONCAT_LAYER && n.crop_h)
        in_e = (int64_t)n.c1 * n.crop_h * n.crop_w + (int64_t)n.c2 * n.in_h * n.in_w;
      if (n.type == ATTENTION_LAYER) {
        const size_t cells = b * (size_t)(n.in_h / 2) * (n.in_w / 2);
        n.z.ensure(sizeof(float) * cells * n.att_ci());  // int_1: the ReLU output of the additive projection
        n.bn.ensure(sizeof(float) * cells);              // int_2: the attention coefficients
        in_e = n.in_elems() + (int64_t)n.c2 * (n.in_h / 2) * (n.in_w / 2);  // gradients of x and g side by side
      }
      max_in = std::max(max_in, in_e);
    }
    scratch_g1_.ensure(sizeof(float) * b * (size_t)max_out);
    scratch_g2_.ensure(sizeof(float) * b * (size_t)max_out);
    scratch_dx_.ensure(sizeof(float) * b * (size_t)max_in);
    in_images_.ensure(sizeof(float) * b * (size_t)in_c_ * in_h_ * in_w_);
    const Node &f = nodes_[final_];
    in_masks_.ensure(sizeof(float) * b * (size_t)f.out_h * f.out_w);
    loss_values_.ensure(sizeof(float) * 3 * b);
    alloc_batch_ = batch;
  }
  // cross-entropy networks: arg max over the final layer's channels of the last forward, [batch][H][W] floats
  void class_map_to_host(float *dst, int batch) {
    const Node &f = nodes_[final_];
    const size_t n = (size_t)batch * f.out_h * f.out_w;
    scratch_g1_.ensure(sizeof(float) * n);
    UNET_CALL(unet_argmax_channels_f32(f.act.f32(), scratch_g1_.f32(), batch, f.out_c, (int64_t)f.out_h * f.out_w, stream_));
    UNET_CALL(unet_memcpy_d2h(dst, scratch_g1_.raw(), sizeof(float) * n, 1, stream_));
  }
  float *input_images_ptr() const { return in_images_.f32(); }
  float *input_masks_ptr() const { return in_masks_.f32(); }

 private:
  void require_created() const {
    if (nodes_.empty()) throw std::runtime_error("create() must be called before using the network");
  }

  void init_node(Node &n, const LayerDesc &d) {
    n.type = d.type;
    n.parents = d.parents;
    n.activation = d.activation;
    n.leaky = (float)d.leaky_const;
    const std::vector<int> &s = d.descriptor;
    auto parent = [&](int i) -> const Node & { return nodes_[n.parents[i]]; };
    switch (d.type) {
      case CONV_LAYER: {
        if (n.parents.size() > 1)
          throw std::runtime_error("Convolution layers cannot have more than one parent");
        if (s.size() != 7)
          throw std::runtime_error(
              "convolution layers need 7 shape values: H_in, W_in, C_in, C_out, kernel, padding, stride");
        n.in_h = s[0]; n.in_w = s[1]; n.in_c = s[2]; n.out_c = s[3];
        n.kernel = s[4]; n.padding = s[5]; n.stride = s[6];
        if (n.kernel < 1 || n.kernel % 2 == 0) throw std::runtime_error("kernel must be square and odd.");
        if (n.in_h < 1 || n.in_w < 1 || n.in_c < 1 || n.out_c < 1 || n.padding < 0 || n.stride < 1)
          throw std::runtime_error("convolution layer: sizes must be positive");
        n.out_h = (n.in_h + 2 * n.padding - n.kernel) / n.stride + 1;
        n.out_w = (n.in_w + 2 * n.padding - n.kernel) / n.stride + 1;
        if (n.out_h < 1 || n.out_w < 1) throw std::runtime_error("convolution layer: empty output");
        n.input_shape = {n.in_c, n.in_h, n.in_w};
        check_parent_shape(n, "convolution");
        n.num_wb = (int64_t)n.kernel * n.kernel * n.in_c * n.out_c + n.out_c;
        n.num_bn = 2 * (int64_t)n.out_c;
        n.fan = (int64_t)n.kernel * n.kernel * n.in_c;
        break;
      }
      case MAX_POOL_LAYER: {
        if (n.parents.size() > 1) {
          std::stringstream ss;
          ss << "max pool layer must have zero or one parent. Recieved " << n.parents.size() << " parents.";
          throw std::runtime_error(ss.str());
        }
        if (s.size() != 5)
          throw std::runtime_error("max pool layers need 5 shape values: C, H, W, kernel, stride");
        n.in_c = s[0]; n.in_h = s[1]; n.in_w = s[2]; n.kernel = s[3]; n.stride = s[4];
        if (n.in_c < 1 || n.in_h < 1 || n.in_w < 1 || n.kernel < 1 || n.stride < 1)
          throw std::runtime_error("max pool layer: sizes must be positive");
        if (n.in_h % n.stride != 0 || n.in_w % n.stride != 0) {
          std::stringstream ss;
          ss << "max pool layer: input " << n.in_h << "x" << n.in_w << " is not divisible by stride " << n.stride;
          throw std::runtime_error(ss.str());
        }
        n.out_c = n.in_c;
        n.out_h = n.in_h / n.stride;
        n.out_w = n.in_w / n.stride;
        n.input_shape = {n.in_c, n.in_h, n.in_w};
        check_parent_shape(n, "max pool");
        break;
      }
      case CONCAT_LAYER: {
        if (n.parents.size() != 2) {
          std::stringstream ss;
          ss << "Concat layer must have two parents. Current parent count is " << n.parents.size();
          throw std::runtime_error(ss.str());
        }
        const Node &a = parent(0), &b = parent(1);
        // A larger first parent (the skip of a U-Net of unpadded convolutions) is centre-cropped to the second
        // one's size when the margins are even; the reference rejects every size mismatch (layer.h:523-533).
        const bool crop = a.out_h >= b.out_h && a.out_w >= b.out_w && (a.out_h != b.out_h || a.out_w != b.out_w) &&
                          (a.out_h - b.out_h) % 2 == 0 && (a.out_w - b.out_w) % 2 == 0;
        if (crop) {
          n.crop_h = a.out_h;
          n.crop_w = a.out_w;
        } else if (a.out_h != b.out_h || a.out_w != b.out_w) {
          std::stringstream ss;
          ss << "Concat layer must have equal heights and widths. Height and width of parent 1: " << a.out_h
             << ", " << a.out_w << " Height and width of parent 2: " << b.out_h << ", " << b.out_w;
          throw std::runtime_error(ss.str());
        }
        n.c1 = a.out_c; n.c2 = b.out_c;
        n.in_c = n.c1 + n.c2; n.in_h = b.out_h; n.in_w = b.out_w;
        n.out_c = n.in_c; n.out_h = n.in_h; n.out_w = n.in_w;
        n.input_shape = {n.c1, n.c2, n.in_h, n.in_w};
        break;
      }
      case UPSAMPLING_LAYER: {
        if (n.parents.size() > 1) {
          std::stringstream ss;
          ss << "upsampling layers need 0 or 1 parents: " << n.parents.size() << " parents givent";
          throw std::runtime_error(ss.str());
        }
        if (s.size() != 5)
          throw std::runtime_error("upsampling layers need 5 shape values: H_in, W_in, C_in, C_out, scale");
        n.in_h = s[0]; n.in_w = s[1]; n.in_c = s[2]; n.out_c = s[3]; n.kernel = s[4];
        if (n.in_c < 1 || n.in_h < 1 || n.in_w < 1 || n.out_c < 1 || n.kernel < 1)
          throw std::runtime_error("upsampling layer: sizes must be positive");
        n.stride = n.kernel;
        n.out_h = n.kernel * n.in_h;
        n.out_w = n.kernel * n.in_w;
        n.input_shape = {n.in_c, n.in_h, n.in_w};
        if (!n.parents.empty()) {
          const Node &p = parent(0);
          if (p.out_c != n.in_c || p.out_h != n.in_h || p.out_w != n.in_w) {
            std::stringstream ss;
            ss << "input shape to upsampling layer does not equal output shape of parent: " << std::endl
               << "upsampling input: " << n.in_c << ", " << n.in_h << ", " << n.in_w << ", " << std::endl
               << "parent output: " << p.out_c << ", " << p.out_h << ", " << p.out_w << ", ";
            throw std::runtime_error(ss.str());
          }
        }
        n.num_wb = (int64_t)n.in_c * n.out_c * n.kernel * n.kernel + n.out_c;
        n.num_bn = 2 * (int64_t)n.out_c;
        n.fan = (int64_t)n.in_c;  // each output pixel sums C_in products
        break;
      }
      case ATTENTION_LAYER: {
        // layer.h:521-571: parents = (x, g), g half the size of x; the descriptor carries no shape
        if (n.parents.size() != 2) {
          std::stringstream ss;
          ss << "Attention layers need two parents: " << n.parents.size() << " parents givent";
          throw std::runtime_error(ss.str());
        }
        const Node &x = parent(0), &g = parent(1);
        if (g.out_h != x.out_h / 2 || g.out_w != x.out_w / 2 || x.out_h % 2 != 0 || x.out_w % 2 != 0) {
          std::stringstream ss;
          ss << "attention not implemented for g that is not half the size of x. g dimensions: " << g.out_h << ", "
             << g.out_w << std::endl
             << "x dimensions: " << x.out_h << ", " << x.out_w;
          throw std::runtime_error(ss.str());
        }
        if (x.out_c < 2) throw std::runtime_error("attention layer: x needs at least two channels");
        n.c1 = x.out_c; n.c2 = g.out_c;
        n.in_c = x.out_c; n.in_h = x.out_h; n.in_w = x.out_w;
        n.out_c = n.in_c; n.out_h = n.in_h; n.out_w = n.in_w;
        n.input_shape = {x.out_c, x.out_h, x.out_w, g.out_c, g.out_h, g.out_w};
        n.num_wb = unet_attention_param_count(n.c1, n.c2, n.att_ci());
        n.num_bn = 0;
        n.fan = (int64_t)n.c1 + n.c2;
        break;
      }
      case BN_LAYER: {
        // A batch norm of its own (the reference declares the type and leaves it empty, layer.h:58-59).
        // Descriptor: [] (shape from the parent) or [C, H, W]; parameters [gamma0, beta0, ...] like the batch
        // norm inside CONV layers; the descriptor's activation follows (NONE = none: there is no reference
        // behaviour to keep for this type).
        if (n.parents.size() != 1) throw std::runtime_error("BN layers need exactly one parent");
        if (!s.empty() && s.size() != 3) throw std::runtime_error("BN layers take no shape values, or 3: C, H, W");
        const Node &p = parent(0);
        n.in_c = s.empty() ? p.out_c : s[0];
        n.in_h = s.empty() ? p.out_h : s[1];
        n.in_w = s.empty() ? p.out_w : s[2];
        check_parent_shape(n, "BN");
        n.out_c = n.in_c; n.out_h = n.in_h; n.out_w = n.in_w;
        n.input_shape = {n.in_c, n.in_h, n.in_w};
        n.num_wb = 0;
        n.num_bn = 2 * (int64_t)n.out_c;
        break;
      }
      case FC_LAYER: {
        // A fully connected layer (declared and left empty in the reference, layer.h:64-65): every output is a
        // weighted sum of the whole [C, H, W] input.  Descriptor: [C_in, H_in, W_in, out_features]; output
        // [out_features, 1, 1]; weights [C_in*H_in*W_in][out_features] then the biases, the layout of a 1x1
        // convolution over a one-pixel image, which is how it runs; no batch norm; activation as for BN layers.
        if (n.parents.size() > 1) throw std::runtime_error("FC layers cannot have more than one parent");
        if (s.size() != 4) throw std::runtime_error("FC layers need 4 shape values: C_in, H_in, W_in, out_features");
        n.in_c = s[0]; n.in_h = s[1]; n.in_w = s[2]; n.out_c = s[3];
        if (n.in_c < 1 || n.in_h < 1 || n.in_w < 1 || n.out_c < 1) throw std::runtime_error("FC layer: sizes must be positive");
        if ((int64_t)n.in_c * n.in_h * n.in_w > (1ll << 24))
          throw std::runtime_error("FC layer: more than 2^24 input features");
        check_parent_shape(n, "FC");
        n.out_h = n.out_w = 1;
        n.input_shape = {n.in_c, n.in_h, n.in_w};
        n.num_wb = n.in_elems() * n.out_c + n.out_c;
        n.num_bn = 0;
        n.fan = n.in_elems();
        break;
      }
      default:
        throw std::runtime_error("unknown layer type");
    }
    n.output_shape = {n.out_c, n.out_h, n.out_w};
  }

  void check_parent_shape(const Node &n, const char *what) {
    if (n.parents.empty()) return;
    const Node &p = nodes_[n.parents[0]];
    if (p.out_c != n.in_c || p.out_h != n.in_h || p.out_w != n.in_w) {
      std::stringstream ss;
      ss << "input shape to " << what << " layer does not equal output shape of parent: layer input: " << n.in_c
         << ", " << n.in_h << ", " << n.in_w << " parent output: " << p.out_c << ", " << p.out_h << ", "
         << p.out_w;
      throw std::runtime_error(ss.str());
    }
  }

  // The activation a CONV / UPSAMPLING layer applies.  SIGMOID, TANH and LEAKY_ReLU are what they say (on
  // convolution layers); every other value -- NONE, the LayerDesc default that main.py:187-224 relies on,
  // included -- is ReLU, as in the reference (convolution_layer_functions.h:59-65,
  // upsampling_layer_functions.h:57-65).
  // Under the cross-entropy loss the final layer's SOFTMAX is the identity here (returned as NONE): the
  // softmax is part of the loss.
  // BN and FC layers, which the reference never built, take their descriptor's activation literally (NONE = none,
  // SOFTMAX = the logits of the cross-entropy loss).  UNET_NONE_IS_IDENTITY=1 does the same for NONE on CONV and
  // UPSAMPLING layers: an opt-in, because a graph written for the reference trains with ReLU there.
  Activation_type effective_activation(const Node &n) const {
    if ((n.type == CONV_LAYER || n.type == BN_LAYER || n.type == FC_LAYER) &&
        (n.activation == SIGMOID || n.activation == TANH || n.activation == LEAKY_ReLU))
      return n.activation;
    if (loss_ == LossFunction::kCrossEntropyLoss && &n == &nodes_[final_]) return NONE;
    if (n.type == BN_LAYER || n.type == FC_LAYER) return n.activation == ReLU ? ReLU : NONE;
    if (n.activation == NONE && none_is_identity_) return NONE;
    return ReLU;
  }
  // the same as the activation code of the fused bf16 batch-norm kernels: 0 ReLU, 1 sigmoid, 3 leaky ReLU, 4 tanh
  int act_code(const Node &n) const {
    switch (effective_activation(n)) {
      case SIGMOID: return 1;
      case LEAKY_ReLU: return 3;
      case TANH: return 4;
      default: return 0;
    }
  }
  void apply_activation(Node &n, int64_t out_n) {  // n.bn -> n.act
    switch (effective_activation(n)) {
      case SIGMOID: UNET_CALL(unet_sigmoid_f32(n.bn.f32(), n.act.f32(), out_n, stream_)); break;
      case TANH: UNET_CALL(unet_tanh_f32(n.bn.f32(), n.act.f32(), out_n, stream_)); break;
      case LEAKY_ReLU: UNET_CALL(unet_leaky_relu_f32(n.bn.f32(), n.act.f32(), out_n, n.leaky, stream_)); break;
      case NONE: UNET_CALL(unet_memcpy_d2d(n.act.raw(), n.bn.raw(), sizeof(float) * (size_t)out_n, stream_)); break;
      default: UNET_CALL(unet_relu_f32(n.bn.f32(), n.act.f32(), out_n, stream_));
    }
  }
  // gradient with respect to the activation's argument: n.grad_out -> the returned pointer (g2, or grad_out itself)
  float *activation_backward(Node &n, float *g2, int64_t out_n) {
    switch (effective_activation(n)) {
      case SIGMOID: UNET_CALL(unet_sigmoid_backward_f32(n.grad_out.f32(), n.act.f32(), g2, out_n, stream_)); break;
      case TANH: UNET_CALL(unet_tanh_backward_f32(n.grad_out.f32(), n.act.f32(), g2, out_n, stream_)); break;
      case LEAKY_ReLU:
        UNET_CALL(unet_leaky_relu_backward_f32(n.grad_out.f32(), n.bn.f32(), g2, out_n, n.leaky, stream_));
        break;
      case NONE: return n.grad_out.f32();
      default: UNET_CALL(unet_relu_backward_f32(n.grad_out.f32(), n.bn.f32(), g2, out_n, stream_));
    }
    return g2;
  }

  const float *layer_input(const Node &n, const float *images) const {
    return n.parents.empty() ? images : nodes_[n.parents[0]].act.f32();
  }

  // conv -> BN -> ReLU|Sigmoid ; tconv -> BN -> ReLU  (convolution_layer_functions.h:6-72,
  // upsampling_layer_functions.h:9-70; BN always uses the layer's own gamma/beta)
  void forward_node(Node &n, const float *images, int batch, bool inference) {
    const int64_t out_n = (int64_t)batch * n.out_elems();
    switch (n.type) {
      case CONV_LAYER:
      case UPSAMPLING_LAYER: {
        const float *in = layer_input(n, images);
        const float *w = params_.f32() + n.param_off;
        const float *gb = w + n.num_wb;
        if (n.type == CONV_LAYER)
          UNET_CALL(unet_convolve_f32(w, in, n.z.f32(), out_n, batch, n.kernel, n.in_c, n.out_c, n.out_h, n.out_w,
                                      n.in_h, n.in_w, n.padding, n.stride, stream_));
        else
          UNET_CALL(unet_upsample_f32(in, w, n.z.f32(), n.kernel, n.in_c, n.out_c, n.in_h, n.in_w, batch, stream_));
        if (inference && effective_activation(n) == ReLU) {  // one pass instead of two; the same values
          UNET_CALL(unet_batch_norm_inference_relu_f32(n.z.f32(), n.act.f32(), gb, n.running.f32(), n.out_h, n.out_w,
                                                       n.out_c, batch, stream_));
          break;
        }
        if (!inference && effective_activation(n) == ReLU) {
          // batch norm and ReLU in one pass; backward takes the ReLU mask from the output (the pre-activation
          // tensor is never written)
          UNET_CALL(unet_batch_norm_train_relu_f32(n.z.f32(), n.act.f32(), gb, n.batch_stats.f32(), n.running.f32(),
                                                   n.out_c, n.out_h, n.out_w, batch, stream_));
          break;
        }
        if (inference)
          UNET_CALL(unet_batch_norm_inference_f32(n.z.f32(), n.bn.f32(), gb, n.running.f32(), n.out_h, n.out_w,
                                                  n.out_c, batch, stream_));
        else
          UNET_CALL(unet_batch_norm_train_f32(n.z.f32(), n.bn.f32(), gb, n.batch_stats.f32(), n.running.f32(),
                                              n.out_c, n.out_h, n.out_w, batch, stream_));
        apply_activation(n, out_n);
        break;
      }
      case BN_LAYER: {
        const float *gb = params_.f32() + n.param_off;
        const float *in = layer_input(n, images);
        if (inference)
          UNET_CALL(unet_batch_norm_inference_f32(in, n.bn.f32(), gb, n.running.f32(), n.out_h, n.out_w, n.out_c, batch,
                                                  stream_));
        else
          UNET_CALL(unet_batch_norm_train_f32(in, n.bn.f32(), gb, n.batch_stats.f32(), n.running.f32(), n.out_c,
                                              n.out_h, n.out_w, batch, stream_));
        apply_activation(n, out_n);
        break;
      }
      case FC_LAYER:  // a 1x1 convolution over a one-pixel image with C_in * H_in * W_in channels
        UNET_CALL(unet_convolve_f32(params_.f32() + n.param_off, layer_input(n, images), n.bn.f32(), out_n, batch, 1,
                                    (int)n.in_elems(), n.out_c, 1, 1, 1, 1, 0, 1, stream_));
        apply_activation(n, out_n);
        break;
      case MAX_POOL_LAYER:
        UNET_CALL(unet_max_pool_f32(layer_input(n, images), n.act.f32(), batch, n.stride, n.kernel, n.in_c, n.in_h,
                                    n.in_w, stream_));
        break;
      case CONCAT_LAYER:
        if (n.crop_h)
          UNET_CALL(unet_concat_crop_f32(nodes_[n.parents[0]].act.f32(), nodes_[n.parents[1]].act.f32(), n.act.f32(),
                                         batch, n.c1, n.c2, n.crop_h, n.crop_w, n.in_h, n.in_w, stream_));
        else
          UNET_CALL(unet_concat_f32(nodes_[n.parents[0]].act.f32(), nodes_[n.parents[1]].act.f32(), n.act.f32(), batch,
                                    (int64_t)n.in_h * n.in_w, n.c1, n.c2, stream_));
        break;
      case ATTENTION_LAYER:  // attention_layer_functions.h:4-28
        UNET_CALL(unet_attention_f32(n.z.f32(), n.bn.f32(), n.act.f32(), params_.f32() + n.param_off,
                                     nodes_[n.parents[0]].act.f32(), nodes_[n.parents[1]].act.f32(), batch, n.in_h,
                                     n.in_w, n.c1, n.c2, n.att_ci(), stream_));
        break;
      default:
        throw std::runtime_error("unsupported layer type in forward");
    }
  }

  // Where a layer's input gradient goes: straight into the parent's grad_out
  // when it is the parent's first contribution, else into scratch followed by
  // an add (layer.h:150-165 does a copy or cudaAddBtoA).
  float *grad_target(Node &parent) { return parent.grad_seen ? scratch_dx_.f32() : parent.grad_out.f32(); }
  void grad_commit(Node &parent, const float *written, int64_t count) {
    if (parent.grad_seen)
      UNET_CALL(unet_add_b_to_a_f32(parent.grad_out.f32(), written, count, stream_));
    parent.grad_seen = true;
  }

  void backward_node(Node &n, const float *images, int batch) {
    if (!n.grad_seen) return;  // not on a path to the loss
    const int64_t out_n = (int64_t)batch * n.out_elems();
    const int64_t in_n = (int64_t)batch * n.in_elems();
    switch (n.type) {
      case CONV_LAYER:
      case UPSAMPLING_LAYER: {
        const float *in = layer_input(n, images);
        const float *w = params_.f32() + n.param_off;
        float *gw = grads_.f32() + n.param_off;
        float *g1 = scratch_g1_.f32();
        if (effective_activation(n) == ReLU) {  // ReLU backward folded into the batch-norm backward passes
          UNET_CALL(unet_batch_norm_relu_backward_f32(n.grad_out.f32(), n.act.f32(), n.z.f32(), n.batch_stats.f32(),
                                                      w + n.num_wb, g1, gw + n.num_wb, n.out_c, n.out_h, n.out_w, batch,
                                                      stream_));
        } else {
          const float *g2 = activation_backward(n, scratch_g2_.f32(), out_n);
          UNET_CALL(unet_batch_norm_backward_f32(g2, n.z.f32(), n.batch_stats.f32(), w + n.num_wb, g1, gw + n.num_wb,
                                                 n.out_c, n.out_h, n.out_w, batch, stream_));
        }
        float *dx = nullptr;
        Node *p = n.parents.empty() ? nullptr : &nodes_[n.parents[0]];
        if (p) dx = grad_target(*p);
        if (n.type == CONV_LAYER)
          UNET_CALL(unet_convolve_backward_f32(g1, in, w, dx, gw, batch, n.kernel, n.in_c, n.out_c, n.out_h, n.out_w,
                                               n.in_h, n.in_w, n.padding, n.stride, stream_));
        else
          UNET_CALL(unet_upsample_backward_f32(g1, in, w, dx, gw, n.kernel, n.in_c, n.out_c, n.in_h, n.in_w, batch,
                                               stream_));
        if (p) grad_commit(*p, dx, in_n);
        break;
      }
      case BN_LAYER: {
        Node &p = nodes_[n.parents[0]];
        const float *g2 = activation_backward(n, scratch_g2_.f32(), out_n);
        float *dx = grad_target(p);
        UNET_CALL(unet_batch_norm_backward_f32(g2, p.act.f32(), n.batch_stats.f32(), params_.f32() + n.param_off, dx,
                                               grads_.f32() + n.param_off, n.out_c, n.out_h, n.out_w, batch, stream_));
        grad_commit(p, dx, in_n);
        break;
      }
      case FC_LAYER: {
        const float *g2 = activation_backward(n, scratch_g2_.f32(), out_n);
        float *dx = nullptr;
        Node *p = n.parents.empty() ? nullptr : &nodes_[n.parents[0]];
        if (p) dx = grad_target(*p);
        UNET_CALL(unet_convolve_backward_f32(g2, layer_input(n, images), params_.f32() + n.param_off, dx,
                                             grads_.f32() + n.param_off, batch, 1, (int)n.in_elems(), n.out_c, 1, 1, 1, 1, 0,
                                             1, stream_));
        if (p) grad_commit(*p, dx, in_n);
        break;
      }
      case MAX_POOL_LAYER: {
        if (n.parents.empty()) break;
        Node &p = nodes_[n.parents[0]];
        float *dx = grad_target(p);
        UNET_CALL(unet_max_pool_backward_f32(n.grad_out.f32(), p.act.f32(), n.act.f32(), dx, batch, n.stride, n.in_c,
                                             n.in_h, n.in_w, stream_));
        grad_commit(p, dx, in_n);
        break;
      }
      case CONCAT_LAYER: {
        // true split: [N,C1,H,W] to parent 0, [N,C2,H,W] to parent 1
        float *tmp = scratch_dx_.f32();
        const int64_t n1 = (int64_t)batch * n.c1 * (n.crop_h ? (int64_t)n.crop_h * n.crop_w : (int64_t)n.in_h * n.in_w);
        const int64_t n2 = (int64_t)batch * n.c2 * n.in_h * n.in_w;
        if (n.crop_h)  // the skip tensor's gradient is zero outside the cropped window
          UNET_CALL(unet_concat_crop_backward_f32(n.grad_out.f32(), tmp, tmp + n1, batch, n.c1, n.c2, n.crop_h, n.crop_w,
                                                  n.in_h, n.in_w, stream_));
        else
          UNET_CALL(unet_concat_backward_f32(n.grad_out.f32(), tmp, n.in_h, n.in_w, n.c1, n.c2, batch, stream_));
        Node &a = nodes_[n.parents[0]], &b = nodes_[n.parents[1]];
        commit_copy(a, tmp, n1);
        commit_copy(b, tmp + n1, n2);
        break;
      }
      case ATTENTION_LAYER: {  // attention_layer_functions.h:30-58: dLdX holds the x gradient, then the g gradient
        Node &x = nodes_[n.parents[0]], &g = nodes_[n.parents[1]];
        const int64_t n1 = (int64_t)batch * x.out_elems(), n2 = (int64_t)batch * g.out_elems();
        float *tmp = scratch_dx_.f32();
        UNET_CALL(unet_attention_backward_f32(n.grad_out.f32(), n.z.f32(), n.bn.f32(), x.act.f32(), g.act.f32(),
                                              params_.f32() + n.param_off, scratch_g1_.f32(), scratch_g2_.f32(), tmp,
                                              tmp + n1, grads_.f32() + n.param_off, batch, n.in_h, n.in_w, n.c1, n.c2,
                                              n.att_ci(), stream_));
        commit_copy(x, tmp, n1);
        commit_copy(g, tmp + n1, n2);
        break;
      }
      default:
        throw std::runtime_error("unsupported layer type in backward");
    }
  }
  void commit_copy(Node &parent, const float *src, int64_t count) {
    if (parent.grad_seen)
      UNET_CALL(unet_add_b_to_a_f32(parent.grad_out.f32(), src, count, stream_));
    else
      UNET_CALL(unet_memcpy_d2d(parent.grad_out.raw(), src, sizeof(float) * (size_t)count, stream_));
    parent.grad_seen = true;
  }

  void record_finished_buckets(int layer_done) {
    const std::vector<Bucket> &b = buckets();
    while (next_bucket_ < b.size() && b[next_bucket_].first_layer >= layer_done) {
      UNET_CALL(unet_event_record(bucket_events_[next_bucket_], stream_));
      // data-parallel training: the bucket's mean over the ranks starts on the communication stream as soon
      // as its last gradient is written, while backward goes on with the earlier layers
      if (comm_)
        UNET_CALL(unet_allreduce_begin(comm_, grads_.f32() + b[next_bucket_].offset, b[next_bucket_].count,
                                       bucket_events_[next_bucket_]));
      ++next_bucket_;
    }
  }

  // ======================= bf16 / NHWC compute path ============================
  static int pad32(int c) { return (c + 31) / 32 * 32; }
  // rows of partial sums the 3x3 convolution entry points write for a call of this shape
  static int conv_stat_rows(int batch, int h, int w, int cip, int cop) {
    return unet_conv3x3_stat_rows(batch, h, w, cip, cop);
  }

  // Decides whether the graph can run on the bf16 kernels and, if so, lays out
  // the activations: a layer feeding a CONCAT writes straight into its slice of
  // the concat buffer, so concat forward and backward move no data.
  bool plan_bf16(std::string &why) {
    if (none_is_identity_)
      for (size_t i = 0; i < nodes_.size(); ++i)
        if (nodes_[i].has_params() && nodes_[i].activation == NONE &&
            !((int)i == final_ && loss_ == LossFunction::kCrossEntropyLoss))
          return why = "a layer without activation (UNET_NONE_IS_IDENTITY)", false;
    for (Node &n : nodes_) {
      n.concat_child = -1;
      n.pool_child = -1;
      n.narrow = false;
      n.head = false;
    }
    std::vector<int> nchild(nodes_.size(), 0);
    for (size_t i = 0; i < nodes_.size(); ++i) {
      Node &n = nodes_[i];
      for (size_t k = 0; k < n.parents.size(); ++k) {
        Node &p = nodes_[n.parents[k]];
        nchild[n.parents[k]]++;
        if (n.type == CONCAT_LAYER) {
          if (p.concat_child >= 0) return why = "a layer feeds two concat layers", false;
          if (n.parents[0] == n.parents[1]) return why = "a concat of a layer with itself", false;
          p.concat_child = (int)i;
          p.concat_slot = (int)k;
        }
      }
    }
    for (size_t i = 0; i < nodes_.size(); ++i) {
      Node &n = nodes_[i];
      const bool is_final = (int)i == final_;
      switch (n.type) {
        case CONV_LAYER:
          if (n.kernel != 3 || n.padding != 1 || n.stride != 1) return why = "a convolution that is not 3x3/pad 1/stride 1", false;
          if (n.parents.empty() ? n.in_c > 32 : n.in_c % 32 != 0) return why = "a convolution whose C_in is not a multiple of 32", false;
          if (n.out_c % 32 != 0) {
            if (!is_final || n.out_c > 32) return why = "a hidden convolution whose C_out is not a multiple of 32", false;
            n.narrow = true;
          }
          // cross-entropy reads fp32 logits [N][C][H][W]: the final layer takes the narrow route (its batch norm
          // and the loss on the fp32 kernels) whatever its channel count, up to one 32-channel block
          if (is_final && loss_ == LossFunction::kCrossEntropyLoss) {
            if (n.out_c > 32) return why = "a cross-entropy loss over more than 32 classes", false;
            n.narrow = true;
          }
          break;
        case UPSAMPLING_LAYER:
          if (n.kernel != 2 || n.parents.empty() || n.in_c % 32 || n.out_c % 32)
            return why = "an upsampling layer that is not scale 2 with channels in multiples of 32", false;
          break;
        case MAX_POOL_LAYER:
          if (n.kernel != 2 || n.stride != 2 || n.parents.empty() || n.in_c % 32 || nchild[i] != 1 || n.concat_child >= 0)
            return why = "a max pool that is not 2x2/stride 2 with one consumer", false;
          break;
        case CONCAT_LAYER:
          if (n.c1 % 32 || n.c2 % 32 || n.concat_child >= 0 || is_final) return why = "a concat with channel counts off multiples of 32", false;
          if (n.crop_h) return why = "a concat that crops its first parent", false;
          break;
        case ATTENTION_LAYER:
          // the gate's projections run in fp32 inside unet_attention_bf16; its tensors stay NHWC bf16
          if (n.c1 % 32 || n.c2 % 32 || is_final) return why = "an attention gate with channel counts off multiples of 32", false;
          break;
        default:
          return why = "an unsupported layer type", false;
      }
      if (is_final && n.type != CONV_LAYER) return why = "a final layer that is not a convolution", false;
      if (n.parents.empty() && n.type != CONV_LAYER) return why = "an input layer that is not a convolution", false;
      n.cip = n.parents.empty() ? 32 : pad32(n.in_c);
      n.cop = pad32(n.out_c);
      n.head = n.narrow && n.type == CONV_LAYER && n.out_c == 1 && !n.parents.empty() &&
               unet_conv3x3_head_supported(n.cip) != 0;
    }
    for (size_t i = 0; i < nodes_.size(); ++i) {
      if (nodes_[i].type != MAX_POOL_LAYER) continue;
      Node &p = nodes_[nodes_[i].parents[0]];
      if (p.has_params() && !p.narrow && p.pool_child < 0 && act_code(p) <= 1) p.pool_child = (int)i;  // (the pooled kernels know ReLU and sigmoid)
    }
    return true;
  }

  void ensure_batch_bf16(int batch) {
    const size_t b = (size_t)batch;
    size_t max_dz = 0;
    for (Node &n : nodes_) {
      const size_t pix = b * (size_t)n.out_h * n.out_w;
      if (n.has_params()) {
        n.zb.ensure(2 * pix * n.cop);
        n.coef.ensure(sizeof(float) * (size_t)unet_bn_coef_floats(n.cop));
        n.coef_inf.ensure(sizeof(float) * (size_t)unet_bn_coef_floats(n.cop));
        const size_t in_pix_tiles = (size_t)unet_conv3x3_stat_tiles(n.in_h, n.in_w);
        // (a convolution's row count follows the kernel that serves the call, which a later change of the
        // pipelined kernel's work threshold can switch: room for either kernel's rows)
        const size_t parts = n.type == CONV_LAYER
                                 ? std::max<size_t>({(size_t)conv_stat_rows(batch, n.in_h, n.in_w, n.cip, n.cop), b * in_pix_tiles,
                                                     std::min<size_t>(4 * b * in_pix_tiles, 1024)})
                                 : 4 * b * in_pix_tiles;
        n.stat_partials.ensure(sizeof(float) * parts * n.cop * 2);
        // BN-backward sums arrive from the consumer's data-gradient kernel (a 3x3 convolution over this
        // layer's output grid, a transposed convolution whose input grid it is, or the head): room for the
        // larger of the row counts those kernels use
        // (the pipelined 3x3 kernel writes four rows per workgroup, with at most 256 workgroups and at most one
        // per 16-row tile -- its smallest tile -- whatever tile height it picks for the shape)
        const size_t tiles16 = b * (size_t)unet_conv3x3_stat_tiles(n.out_h, n.out_w);
        const size_t bwd_rows = std::max<size_t>(tiles16, std::min<size_t>(4 * tiles16, 1024));
        n.bwd_partials.ensure(sizeof(float) * bwd_rows * n.cop * 2);
        max_dz = std::max(max_dz, 2 * pix * n.cop);
        const size_t taps = n.type == CONV_LAYER ? 9 : 4;
        n.w_fwd.ensure(2 * taps * (size_t)n.cop * n.cip);
        n.w_bwd.ensure(2 * taps * (size_t)n.cop * n.cip);
      }
      if (n.type == MAX_POOL_LAYER) n.mask.ensure(pix * n.cop);
      if (n.type == ATTENTION_LAYER) {
        const size_t cells = b * (size_t)(n.in_h / 2) * (n.in_w / 2);
        n.z.ensure(sizeof(float) * cells * n.att_ci());  // int_1
        n.bn.ensure(sizeof(float) * cells);              // int_2 (the coefficients)
      }
      if (n.narrow) {  // fp32 side buffers of the final layer
        const size_t fb = sizeof(float) * b * (size_t)n.out_elems();
        n.z.ensure(fb);
        n.bn.ensure(fb);
        n.act.ensure(fb);
        n.grad_out.ensure(fb);
      } else if (n.concat_child < 0) {
        n.yb.ensure(2 * pix * n.cop);
      }
      if (!n.narrow) n.gb.ensure(2 * pix * n.cop);
    }
    // activation placement (after the concat buffers exist)
    for (Node &n : nodes_) {
      if (n.narrow) continue;
      if (n.concat_child >= 0) {
        Node &c = nodes_[n.concat_child];
        n.y_ptr = c.yb.u16() + (n.concat_slot == 0 ? 0 : c.c1);
        n.y_ld = c.cop;
      } else {
        n.y_ptr = n.yb.u16();
        n.y_ld = n.cop;
      }
    }
    Node &f = nodes_[final_];
    scratch_g1_.ensure(sizeof(float) * b * (size_t)f.out_elems());
    scratch_g2_.ensure(sizeof(float) * b * (size_t)f.out_elems());
    dz16_.ensure(max_dz);
    x16_.ensure(2 * b * (size_t)in_h_ * in_w_ * 32);
    in_images_.ensure(sizeof(float) * b * (size_t)in_c_ * in_h_ * in_w_);
    in_masks_.ensure(sizeof(float) * b * (size_t)f.out_h * f.out_w);
    loss_values_.ensure(sizeof(float) * 3 * b);
    params_dirty_ = true;
    inf_coef_valid_ = false;
    pack_boxes_ = 0;
  }

  // fp32 master weights [C_in][C_out][k][k] -> packed bf16 operands
  // One launch for all layers; the table of per-layer arguments lives on the device and is rebuilt
  // whenever buffers may have moved (ensure_batch_bf16 clears pack_boxes_).
  void pack_all_weights() {
    if (pack_boxes_ == 0) {
      std::vector<int64_t> table;
      int64_t boxes = 0;
      pack_layers_ = 0;
      for (Node &n : nodes_)
        if (n.has_params()) {
          const int64_t row[10] = {(int64_t)(uintptr_t)(params_.f32() + n.param_off), (int64_t)(uintptr_t)n.w_fwd.u16(),
                                   (int64_t)(uintptr_t)n.w_bwd.u16(), n.in_c, n.out_c, n.cip, n.cop,
                                   n.type == CONV_LAYER ? 3 : 2, boxes, 0};
          table.insert(table.end(), row, row + 10);
          boxes += (int64_t)(n.cip / 32) * (n.cop / 32);
          ++pack_layers_;
        }
      if (pack_layers_ == 0) { params_dirty_ = false; return; }
      pack_table_.ensure(sizeof(int64_t) * table.size());
      UNET_CALL(unet_memcpy_h2d(pack_table_.raw(), table.data(), sizeof(int64_t) * table.size(), stream_));
      UNET_CALL(unet_stream_synchronize(stream_));  // `table` is a local
      pack_boxes_ = (int)boxes;
    }
    UNET_CALL(unet_pack_weights_batch_bf16(reinterpret_cast<const int64_t *>(pack_table_.raw()), pack_layers_, pack_boxes_,
                                           stream_));
    params_dirty_ = false;
  }

  void forward_node_bf16(Node &n, int batch, bool inference) {
    const int64_t out_pix = (int64_t)batch * n.out_h * n.out_w;
    switch (n.type) {
      case CONV_LAYER:
      case UPSAMPLING_LAYER: {
        const uint16_t *in = n.parents.empty() ? x16_.u16() : nodes_[n.parents[0]].y_ptr;
        const int in_ld = n.parents.empty() ? in_cp() : nodes_[n.parents[0]].y_ld;
        const int in_cip = n.parents.empty() ? in_cp() : n.cip;  // channels present in memory
        const float *w = params_.f32() + n.param_off;
        const float *bias = w + (n.num_wb - n.out_c);
        const float *gb = w + n.num_wb;
        float *parts = (inference || n.narrow) ? nullptr : n.stat_partials.f32();
        int num_parts;
        const bool fold_bn = inference && !n.narrow && act_code(n) == 0;
        if (fold_bn) {
          // inference: BN (running statistics) + ReLU folded into the conv epilogue, straight into y
          if (n.type == CONV_LAYER)
            UNET_CALL(unet_conv3x3_bn_relu_bf16(in, in_ld, n.w_fwd.u16(), bias, n.out_c, n.coef_inf.f32(), n.y_ptr, n.y_ld, batch,
                                                n.in_h, n.in_w, in_cip, n.cop, stream_));
          else
            UNET_CALL(unet_tconv2x2_bn_relu_bf16(in, in_ld, n.w_fwd.u16(), bias, n.out_c, n.coef_inf.f32(), n.y_ptr, n.y_ld, batch,
                                                 n.in_h, n.in_w, n.cip, n.cop, stream_));
          break;
        }
        if (n.head) {
          UNET_CALL(unet_conv3x3_head_bf16(in, in_ld, n.w_fwd.u16(), bias, n.z.f32(), batch, n.in_h, n.in_w, n.cip, n.cop,
                                           stream_));
          num_parts = 0;
        } else if (n.type == CONV_LAYER) {
          UNET_CALL(unet_conv3x3_bf16(in, in_ld, n.w_fwd.u16(), bias, n.out_c, n.zb.u16(), n.cop, parts, batch, n.in_h,
                                      n.in_w, in_cip, n.cop, 0, stream_));
          num_parts = conv_stat_rows(batch, n.in_h, n.in_w, n.cip, n.cop);
        } else {
          UNET_CALL(unet_tconv2x2_bf16(in, in_ld, n.w_fwd.u16(), bias, n.out_c, n.zb.u16(), n.cop, parts, batch, n.in_h,
                                       n.in_w, n.cip, n.cop, stream_));
          num_parts = 4 * batch * unet_conv3x3_stat_tiles(n.in_h, n.in_w);
        }
        if (n.narrow) {
          // final layer: BN + activation (+ loss) on the fp32 kernels, [N,C,H,W] with C = out_c
          if (!n.head)
            UNET_CALL(unet_nhwc_bf16_to_nchw_f32(n.zb.u16(), n.z.f32(), batch, n.out_c, n.out_h, n.out_w, n.cop, stream_));
          if (inference)
            UNET_CALL(unet_batch_norm_inference_f32(n.z.f32(), n.bn.f32(), gb, n.running.f32(), n.out_h, n.out_w,
                                                    n.out_c, batch, stream_));
          else
            UNET_CALL(unet_batch_norm_train_f32(n.z.f32(), n.bn.f32(), gb, n.batch_stats.f32(), n.running.f32(),
                                                n.out_c, n.out_h, n.out_w, batch, stream_));
          const int64_t out_n = (int64_t)batch * n.out_elems();
          apply_activation(n, out_n);
          break;
        }
        if (!inference)
          UNET_CALL(unet_bn_finalize_bf16(parts, num_parts, n.out_c, n.cop, out_pix, gb, n.batch_stats.f32(),
                                          n.running.f32(), n.coef.f32(), stream_));
        const float *coef = inference ? n.coef_inf.f32() : n.coef.f32();
        const int act = act_code(n);
        if (n.pool_child >= 0) {
          // the skip layers of the encoder: activation, pooled activation and tie mask from one pass over z
          Node &q = nodes_[n.pool_child];
          UNET_CALL(unet_bn_act_pool_bf16(n.zb.u16(), n.cop, n.y_ptr, n.y_ld, coef, q.y_ptr, q.y_ld, q.mask.u8(),
                                          batch, n.out_h, n.out_w, n.cop, act, stream_));
          q.pooled_by_parent = true;
          break;
        }
        UNET_CALL(unet_bn_act_ex_bf16(n.zb.u16(), n.cop, n.y_ptr, n.y_ld, coef, n.cop, out_pix, act, n.leaky, stream_));
        break;
      }
      case MAX_POOL_LAYER: {
        if (n.pooled_by_parent) {
          n.pooled_by_parent = false;
          break;
        }
        const Node &p = nodes_[n.parents[0]];
        UNET_CALL(unet_max_pool2x2_bf16(p.y_ptr, p.y_ld, n.y_ptr, n.y_ld, n.mask.u8(), batch, n.in_h, n.in_w, n.cop,
                                        stream_));
        break;
      }
      case CONCAT_LAYER:
        break;  // both parents already wrote their slices of this layer's buffer
      case ATTENTION_LAYER: {
        // x, g and the output stay NHWC bf16 (the output may be a slice of a concat buffer); the projections
        // run in fp32 inside the call
        const Node &x = nodes_[n.parents[0]], &g = nodes_[n.parents[1]];
        UNET_CALL(unet_attention_bf16(x.y_ptr, x.y_ld, g.y_ptr, g.y_ld, params_.f32() + n.param_off, n.z.f32(),
                                      n.bn.f32(), n.y_ptr, n.y_ld, batch, n.in_h, n.in_w, n.c1, n.c2, n.att_ci(),
                                      stream_));
        break;
      }
      default:
        throw std::runtime_error("unsupported layer type in forward");
    }
  }

  // concat node c: its second parent is a BN + ReLU layer whose only consumer is c, and the channel
  // split sits on a 64-channel block boundary of the data-gradient kernel
  bool concat_second_is_fusable(const Node &c) const {
    const Node &u = nodes_[c.parents[1]];
    return u.has_params() && !u.narrow && u.num_children == 1 && c.num_children == 1 && c.c1 % 64 == 0 &&
           c.cop % 64 == 0 && u.cop == c.c2 && act_code(u) == 0;
  }

  // The gradient of a convolution bias that feeds a batch norm is identically zero: dz, the batch-norm
  // backward output, sums to zero over the batch (sum dz = c0 * (sum g - M*k1 - k2 * sum xhat), with
  // k1 = sum g / M and sum xhat = 0).  The reference adds up the computed dz (convolution.cu:153-414), which
  // yields rounding noise around zero -- noise that Adam then normalises into steps of the size of the
  // learning rate.  On the bf16 path every bias feeds a batch norm, so the engine leaves those gradient
  // entries at the exact value (the vector is cleared once, when it is allocated) and saves a reduction per
  // layer; the fp32 path and the op-level entry points (grad_bias != NULL) still compute the sum.
  static constexpr float *kNoBiasGrad = nullptr;
  void backward_node_bf16(Node &n, int batch) {
    const int64_t out_pix = (int64_t)batch * n.out_h * n.out_w;
    switch (n.type) {
      case CONV_LAYER:
      case UPSAMPLING_LAYER: {
        const float *w = params_.f32() + n.param_off;
        float *gw = grads_.f32() + n.param_off;
        float *gbias = gw + (n.num_wb - n.out_c);
        uint16_t *dz = dz16_.u16();
        if (n.narrow) {
          // fp32: activation backward, BN backward, bias gradient; then dz -> NHWC bf16 (padded with zeros)
          const int64_t out_n = (int64_t)batch * n.out_elems();
          float *g2 = scratch_g2_.f32(), *g1 = scratch_g1_.f32();
          g2 = activation_backward(n, g2, out_n);  // (NONE: the logits of the cross-entropy loss)
          UNET_CALL(unet_batch_norm_backward_f32(g2, n.z.f32(), n.batch_stats.f32(), w + n.num_wb, g1, gw + n.num_wb,
                                                 n.out_c, n.out_h, n.out_w, batch, stream_));
          if (n.head) {
            // one output channel: data and weight gradients straight from the fp32 dz, on the head kernels
            Node &p = nodes_[n.parents[0]];
            const bool fuse = p.has_params() && !p.narrow && p.num_children == 1 && p.concat_child < 0 && !p.g_written &&
                              act_code(p) == 0;
            UNET_CALL(unet_conv3x3_head_dgrad_bf16(g1, w, n.in_c, p.gb.u16(), p.cop, fuse ? p.zb.u16() : nullptr, p.cop,
                                                   fuse ? p.coef.f32() : nullptr, fuse ? p.bwd_partials.f32() : nullptr,
                                                   batch, n.in_h, n.in_w, n.cip, (!fuse && p.g_written) ? 1 : 0, stream_));
            if (fuse) p.bwd_partials_n = batch * unet_conv3x3_stat_tiles(n.in_h, n.in_w);
            p.g_written = true;
            UNET_CALL(unet_conv3x3_head_wgrad_bf16(g1, p.y_ptr, p.y_ld, gw, gbias, batch, n.in_h, n.in_w, n.cip, n.in_c,
                                                   stream_));
            break;
          }
          UNET_CALL(unet_nchw_f32_to_nhwc_bf16(g1, dz, batch, n.out_c, n.out_h, n.out_w, n.cop, n.cop, stream_));
          // bias gradient = per-channel sum of dz: reuse the BN statistics reduction (sum x lands at [2c])
          UNET_CALL(unet_batch_norm_train_f32(g1, g2, w + n.num_wb, n.coef.f32(), nullptr, n.out_c, n.out_h, n.out_w,
                                              batch, stream_));
          for (int c = 0; c < n.out_c; ++c)
            UNET_CALL(unet_memcpy_d2d(gbias + c, n.coef.f32() + 2 * c, sizeof(float), stream_));
        } else {
          if (n.pool_src && n.g_written && n.g2_ptr) {
            // three gradient sources: scatter the pooled one after all
            const Node &q = *n.pool_src;
            UNET_CALL(unet_max_pool2x2_backward_bf16(q.gb.u16(), q.cop, q.mask.u8(), n.gb.u16(), n.cop, batch, q.in_h,
                                                     q.in_w, q.cop, 1, stream_));
            n.pool_src = nullptr;
          }
          const uint16_t *d1 = n.g_written ? n.gb.u16() : n.g2_ptr;
          const int l1 = n.g_written ? n.cop : n.g2_ld;
          const uint16_t *d2 = (n.g_written && n.g2_ptr) ? n.g2_ptr : nullptr;
          if (!d1 && !n.pool_src) return;  // not on a path to the loss
          const int act = act_code(n);
          if (n.pool_src) {
            // a skip layer: the gradient through the max pool is gathered from the pooled gradient and the tie
            // mask inside the batch-norm backward passes, next to the concat slice d1 (if any)
            const Node &q = *n.pool_src;
            UNET_CALL(unet_bn_act_backward_pooled_bf16(d1, l1, q.gb.u16(), q.cop, q.mask.u8(), batch, n.out_h, n.out_w,
                                                       n.zb.u16(), n.cop, n.coef.f32(), w + n.num_wb, dz, n.cop,
                                                       gw + n.num_wb, kNoBiasGrad, n.out_c, n.cop, act, stream_));
            n.pool_src = nullptr;
          } else if (n.bwd_partials_n > 0)  // the consumer's data-gradient kernel already reduced over the pixels
            UNET_CALL(unet_bn_act_backward_from_partials_bf16(d1, l1, n.zb.u16(), n.cop, n.coef.f32(), w + n.num_wb,
                                                              n.bwd_partials.f32(), n.bwd_partials_n, dz, n.cop,
                                                              gw + n.num_wb, kNoBiasGrad, n.out_c, n.cop, out_pix,
                                                              stream_));
          else
            UNET_CALL(unet_bn_act_backward_ex_bf16(d1, l1, d2, n.g2_ld, n.zb.u16(), n.cop, n.coef.f32(), w + n.num_wb, dz,
                                                   n.cop, gw + n.num_wb, kNoBiasGrad, n.out_c, n.cop, out_pix, act,
                                                   n.leaky, stream_));
          (void)gbias;
        }
        const uint16_t *x = n.parents.empty() ? x16_.u16() : nodes_[n.parents[0]].y_ptr;
        const int x_ld = n.parents.empty() ? in_cp() : nodes_[n.parents[0]].y_ld;
        if (!n.parents.empty()) {
          Node &p = nodes_[n.parents[0]];
          // When this is the only gradient the parent's output receives and the parent is a
          // BN + ReLU layer, the data-gradient kernel also does the pixel reduction of the parent's
          // batch-norm backward.
          // UNET_LAB_BRED_MIN_CHANNELS = c: laboratory switch, no fusion into 3x3 data gradients whose output gradient has
          // fewer than c channels (the parent's reduction then runs as a pass of its own)
          static const int lab_min_c = getenv("UNET_LAB_BRED_MIN_CHANNELS") ? atoi(getenv("UNET_LAB_BRED_MIN_CHANNELS")) : 0;
          const bool fuse = p.has_params() && !p.narrow && p.num_children == 1 && p.concat_child < 0 && !p.g_written &&
                            act_code(p) == 0 && !(n.type == CONV_LAYER && n.cop < lab_min_c);
          if (fuse) {
            if (n.type == CONV_LAYER)
              UNET_CALL(unet_conv3x3_dgrad_bnred_bf16(dz, n.cop, n.w_bwd.u16(), p.gb.u16(), p.cop, p.zb.u16(), p.cop,
                                                      p.coef.f32(), p.bwd_partials.f32(), 0, batch, n.in_h, n.in_w,
                                                      n.cop, n.cip, stream_));
            else
              UNET_CALL(unet_tconv2x2_dgrad_bnred_bf16(dz, n.cop, n.w_bwd.u16(), p.gb.u16(), p.cop, p.zb.u16(), p.cop,
                                                       p.coef.f32(), p.bwd_partials.f32(), batch, n.in_h, n.in_w, n.cip,
                                                       n.cop, stream_));
            p.bwd_partials_n = n.type == CONV_LAYER ? conv_stat_rows(batch, n.in_h, n.in_w, n.cop, n.cip)
                                                    : batch * unet_conv3x3_stat_tiles(n.in_h, n.in_w);
          } else if (n.type == CONV_LAYER && p.type == CONCAT_LAYER && !p.g_written && concat_second_is_fusable(p)) {
            // Input = concat(skip, up): the second half of the gradient is all the gradient the
            // upsampling layer's BN + ReLU output receives, so its pixel reduction is fused here too.
            Node &u = nodes_[p.parents[1]];
            UNET_CALL(unet_conv3x3_dgrad_bnred_bf16(dz, n.cop, n.w_bwd.u16(), p.gb.u16(), p.cop, u.zb.u16(), u.cop,
                                                    u.coef.f32(), u.bwd_partials.f32(), p.c1, batch, n.in_h, n.in_w,
                                                    n.cop, n.cip, stream_));
            u.bwd_partials_n = conv_stat_rows(batch, n.in_h, n.in_w, n.cop, n.cip);
          } else if (n.type == CONV_LAYER) {
            UNET_CALL(unet_conv3x3_bf16(dz, n.cop, n.w_bwd.u16(), nullptr, 0, p.gb.u16(), p.cop, nullptr, batch, n.in_h,
                                        n.in_w, n.cop, n.cip, p.g_written ? 1 : 0, stream_));
          } else {
            UNET_CALL(unet_tconv2x2_dgrad_bf16(dz, n.cop, n.w_bwd.u16(), p.gb.u16(), p.cop, batch, n.in_h, n.in_w, n.cip,
                                               n.cop, p.g_written ? 1 : 0, stream_));
          }
          p.g_written = true;
        }
        if (n.type == CONV_LAYER)
          UNET_CALL(unet_conv3x3_wgrad_bf16(dz, n.cop, x, x_ld, gw, batch, n.in_h, n.in_w,
                                            n.parents.empty() ? in_cp() : n.cip, n.cop, n.in_c, n.out_c, stream_));
        else
          UNET_CALL(unet_tconv2x2_wgrad_bf16(dz, n.cop, x, x_ld, gw, batch, n.in_h, n.in_w, n.cip, n.cop, n.in_c,
                                             n.out_c, stream_));
        break;
      }
      case MAX_POOL_LAYER: {
        if (!n.g_written) return;
        Node &p = nodes_[n.parents[0]];
        if (p.has_params() && !p.narrow && !p.pool_src && act_code(p) <= 1) {
          p.pool_src = &n;  // gathered by the parent's batch-norm backward (ReLU and sigmoid layers)
          break;
        }
        UNET_CALL(unet_max_pool2x2_backward_bf16(n.gb.u16(), n.cop, n.mask.u8(), p.gb.u16(), p.cop, batch, n.in_h, n.in_w,
                                                 n.cop, p.g_written ? 1 : 0, stream_));
        p.g_written = true;
        break;
      }
      case CONCAT_LAYER: {
        if (!n.g_written) return;
        Node &a = nodes_[n.parents[0]], &b = nodes_[n.parents[1]];
        a.g2_ptr = n.gb.u16();
        a.g2_ld = n.cop;
        b.g2_ptr = n.gb.u16() + n.c1;
        b.g2_ld = n.cop;
        break;
      }
      case ATTENTION_LAYER: {
        const uint16_t *d1 = n.g_written ? n.gb.u16() : n.g2_ptr;
        const int l1 = n.g_written ? n.cop : n.g2_ld;
        const uint16_t *d2 = (n.g_written && n.g2_ptr) ? n.g2_ptr : nullptr;
        if (!d1) return;  // not on a path to the loss
        Node &x = nodes_[n.parents[0]], &g = nodes_[n.parents[1]];
        UNET_CALL(unet_attention_backward_bf16(d1, l1, d2, n.g2_ld, x.y_ptr, x.y_ld, g.y_ptr, g.y_ld,
                                               params_.f32() + n.param_off, n.z.f32(), n.bn.f32(), x.gb.u16(), x.cop,
                                               x.g_written ? 1 : 0, g.gb.u16(), g.cop, g.g_written ? 1 : 0,
                                               grads_.f32() + n.param_off, batch, n.in_h, n.in_w, n.c1, n.c2,
                                               n.att_ci(), stream_));
        x.g_written = true;
        g.g_written = true;
        break;
      }
      default:
        throw std::runtime_error("unsupported layer type in backward");
    }
  }

  struct InputSlot {
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    DeviceBuffer images, masks, raw;  // raw: the uint8 batch of the byte-input feed
    unet_event_t copied = nullptr, consumed = nullptr;
    bool in_use = false;
    int batch = 0;
  };
  InputSlot slots_[2];
  unet_stream_t copy_stream_ = nullptr;
  DeviceBuffer norm_coef_;  // [scale(C) | shift(C)] of the byte-input feed
  bool norm_set_ = false;

  std::vector<Node> nodes_;
  LossFunction loss_ = LossFunction::kDiceLoss;
  int final_ = -1;
  int in_c_ = 0, in_h_ = 0, in_w_ = 0;
  // channels of the network input as stored in x16_: 8 for up to 8 image channels (16 bytes per pixel), else 32
  int in_cp() const { return in_c_ <= 8 ? 8 : 32; }
  int64_t total_params_ = 0;
  DeviceBuffer params_, grads_, adam_;
  DeviceBuffer scratch_g1_, scratch_g2_, scratch_dx_;
  DeviceBuffer in_images_, in_masks_, loss_values_;
  DeviceBuffer labels_;  // cross-entropy: the masks as int32 class indices
  int alloc_batch_ = 0;
  int step_count_ = 0;
  uint64_t seed_ = 0;
  bool params_dirty_ = true;
  bool inf_coef_valid_ = false;  // Node::coef_inf match the current gamma/beta and running statistics
  DeviceBuffer pack_table_;  // per-layer arguments of the batched weight packing
  int pack_layers_ = 0, pack_boxes_ = 0;
  bool use_bf16_ = false;
  DeviceBuffer x16_;      // network input as NHWC bf16, channels padded to in_cp()
  DeviceBuffer dz16_;     // scratch: gradient w.r.t. a layer's pre-BN output
  bool verbose_ = true;
  bool check_nan_ = false;
  bool none_is_identity_ = false;  // UNET_NONE_IS_IDENTITY=1: NONE on CONV / UPSAMPLING layers means no activation
  DeviceBuffer nan_flag_;
  unet_stream_t stream_ = nullptr;
  int64_t bucket_bytes_ = 8 << 20;
  std::vector<Bucket> buckets_;
  std::vector<void *> bucket_events_;
  size_t next_bucket_ = 0;
  unet_comm_t comm_ = nullptr;
};

}  // namespace unet
