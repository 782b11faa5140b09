// Python module `neural_net`: same classes, enums, method names and argument
// orders as the reference's src/bindings/net.cpp:26-88, so main.py, test.py and
// test/test_convolution.py drive it unchanged.  Additional methods (parameter
// access, single steps, device-pointer entry points for data-parallel
// training) are additive.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <chrono>

#include "net.h"
#include "selftest.h"

namespace py = pybind11;
using unet::Net;
using farray = py::array_t<float, py::array::c_style | py::array::forcecast>;

namespace {

// ---------------------------------------------------------------- NeuralNet --
class NN {
 public:
  NN() {
    if (std::getenv("UNET_QUIET")) net_.set_verbose(false);
    const char *nan = std::getenv("UNET_CHECK_NAN");
    if (nan && nan[0] && nan[0] != '0') net_.set_check_nan(true);
    if (net_.verbose()) std::cout << "initializing NN" << std::endl << std::flush;
    UNET_CALL(unet_device_reset());
  }

  void create(const std::vector<LayerDesc> &desc, LossFunction loss) { net_.create(desc, loss); }

  // neural_net.h:121-183.  images [N,C,H,W] fp32, masks [N,H,W] fp32.
  void train(farray images, farray masks, float learning_rate, int epochs, int batch_size) {
    const bool v = net_.verbose();
    if (v) std::cout << "getting images and mask pointers" << std::endl;
    py::buffer_info ii = images.request(), mi = masks.request();
    if (ii.ndim != 4) throw std::runtime_error("images must have shape [N, C, H, W]");
    if (mi.ndim != 3) throw std::runtime_error("masks must have shape [N, H, W]");
    const int64_t N = ii.shape[0], C = ii.shape[1], H = ii.shape[2], W = ii.shape[3];
    check_input(C, H, W);
    check_mask(N, mi.shape[0], mi.shape[1], mi.shape[2]);
    if (batch_size < 1) throw std::runtime_error("Batch size must be greater than 0");
    const float *img = static_cast<const float *>(ii.ptr);
    const float *msk = static_cast<const float *>(mi.ptr);
    const unet::Node &f = net_.nodes()[net_.final_layer()];
    const int64_t mask_hw = (int64_t)f.out_h * f.out_w;
    run_epochs(N, batch_size, epochs, learning_rate, [&](int slot, int64_t start, int count) {
      net_.stage_batch_host(slot, img + start * C * H * W, msk + start * mask_hw, count);
    });
  }

  // The epoch loop of train(): batch b+1 is staged (host copy to pinned memory, then an asynchronous
  // transfer on a second stream) while the GPU runs batch b.  stage(slot, first image, count).
  template <typename Stage>
  void run_epochs(int64_t N, int batch_size, int epochs, float learning_rate, Stage stage) {
    const bool v = net_.verbose();
    if (v) std::cout << "initiating gradient parameters" << std::endl;
    net_.ensure_batch(batch_size);
    if (v) std::cout << "done init_gradient_parameters" << std::endl;
    const int64_t nb = (N + batch_size - 1) / batch_size;
    unet::DeviceBuffer epoch_lv;  // per-batch loss sums, read once per epoch
    epoch_lv.ensure(sizeof(float) * 3 * (size_t)batch_size * (size_t)nb);
    std::vector<float> lv(3 * (size_t)batch_size * (size_t)nb);
    for (int epoch = 0; epoch < epochs; ++epoch) {
      if (v) std::cout << "beggining epoch " << epoch << "/" << epochs << std::endl;
      const auto t0 = std::chrono::steady_clock::now();
      stage(0, 0, (int)std::min<int64_t>(batch_size, N));
      for (int64_t b = 0; b < nb; ++b) {
        const int64_t start = b * batch_size;
        const int cur = (int)std::min<int64_t>(batch_size, N - start);
        net_.train_batch_staged((int)(b & 1), learning_rate, true);
        if (b + 1 < nb) {
          const int64_t nstart = start + batch_size;
          stage((int)((b + 1) & 1), nstart, (int)std::min<int64_t>(batch_size, N - nstart));
        }
        UNET_CALL(unet_memcpy_d2d(epoch_lv.f32() + 3 * (size_t)batch_size * b, net_.loss_values_ptr(),
                                  sizeof(float) * 3 * cur, net_.stream()));
      }
      net_.sync();
      const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      net_.raise_if_nonfinite();
      // Dice loss: 1 - mean Dice score, as the reference prints.  Difference loss: the mean over the
      // epoch's batches of 0.5 * mean (y - m)^2 (the reference prints the Dice figures of an unrelated,
      // never-filled buffer there); no Dice score is measured, so none is reported.
      const bool is_dice = net_.loss() == LossFunction::kDiceLoss;
      const unet::Node &f = net_.nodes()[net_.final_layer()];
      double total = 0;
      UNET_CALL(unet_memcpy_d2h(lv.data(), epoch_lv.raw(), sizeof(float) * lv.size(), 1, net_.stream()));
      for (int64_t b = 0; b < nb; ++b) {
        const int cur = (int)std::min<int64_t>(batch_size, N - b * batch_size);
        double batch_sq = 0;
        for (int i = 0; i < cur; ++i) {
          const float *s = &lv[3 * ((size_t)batch_size * b + i)];
          if (is_dice)
            total += 2.0 * s[0] / (s[1] + s[2] + 1e-5);
          else
            batch_sq += s[0];
        }
        if (net_.loss() == LossFunction::kCrossEntropyLoss)
          total += lv[3 * (size_t)batch_size * b];  // the batch's mean cross-entropy
        else if (!is_dice)
          total += 0.5 * batch_sq / ((double)cur * f.out_h * f.out_w);
      }
      const double avg_loss = is_dice ? 1.0 - total / (double)N : total / (double)nb;
      last_epoch_loss_ = (float)avg_loss;
      if (v) {
        std::cout << "ephoch " << epoch << " done." << std::endl;
        std::cout << "average loss: " << avg_loss << std::endl;
        if (is_dice)
          std::cout << "average dice score: " << total / (double)N << std::endl;
        else
          std::cout << "average dice score: n/a (not the Dice loss)" << std::endl;
        std::cout << "images/s: " << (double)N / secs << std::endl;
      }
    }
  }

  // ---- byte-input feed (additive): raw uint8 images [N,H,W,C] and uint8 masks [N,H,W]; the /255 and
  // per-channel z-score of main.py:89-92 run on the device, masks are scaled by 1/255 when they hold 0/255
  // (main.py:97-98)
  typedef py::array_t<uint8_t, py::array::c_style | py::array::forcecast> u8array;
  void set_input_normalization(farray mean, farray stddev) {
    if (mean.size() != stddev.size()) throw std::runtime_error("mean and std must have one entry per channel");
    net_.set_input_normalization(mean.data(), stddev.data(), (int)mean.size());
  }
  static float mask_scale_of(const uint8_t *m, size_t n) {
    for (size_t i = 0; i < n; ++i)
      if (m[i] > 1) return 1.0f / 255.0f;
    return 1.0f;
  }
  void train_u8(u8array images, u8array masks, float learning_rate, int epochs, int batch_size) {
    py::buffer_info ii = images.request(), mi = masks.request();
    if (ii.ndim != 4) throw std::runtime_error("images must have shape [N, H, W, C]");
    if (mi.ndim != 3) throw std::runtime_error("masks must have shape [N, H, W]");
    const int64_t N = ii.shape[0], H = ii.shape[1], W = ii.shape[2], C = ii.shape[3];
    if (N != mi.shape[0] || H != mi.shape[1] || W != mi.shape[2])
      throw std::runtime_error("Mask dimensions do not equal image dimensions");
    check_input(C, H, W);
    if (batch_size < 1) throw std::runtime_error("Batch size must be greater than 0");
    const uint8_t *img = static_cast<const uint8_t *>(ii.ptr), *msk = static_cast<const uint8_t *>(mi.ptr);
    const float ms = mask_scale_of(msk, (size_t)mi.size);
    run_epochs(N, batch_size, epochs, learning_rate, [&](int slot, int64_t start, int count) {
      net_.stage_batch_host_u8(slot, img + start * H * W * C, msk + start * H * W, count, ms);
    });
  }
  py::array_t<float> infer_u8(u8array image) {
    py::buffer_info ii = image.request();
    if (ii.ndim != 3) throw std::runtime_error("Only accepting single images [H, W, C]");
    check_input(ii.shape[2], ii.shape[0], ii.shape[1]);
    const unet::Node &f = net_.nodes()[net_.final_layer()];
    std::vector<uint8_t> no_mask((size_t)f.out_h * f.out_w, 0);
    net_.stage_batch_host_u8(0, static_cast<const uint8_t *>(ii.ptr), no_mask.data(), 1, 1.0f);
    net_.infer_batch_staged(0);
    UNET_CALL(unet_threshold_f32(const_cast<float *>(f.act.f32()), 0.5f, f.out_elems(), net_.stream()));
    py::array_t<float> ret({(py::ssize_t)f.out_c, (py::ssize_t)f.out_h, (py::ssize_t)f.out_w});
    net_.copy_output_to_host(net_.final_layer(), ret.mutable_data(), 1);
    return ret;
  }

  py::array_t<float> infer(farray image) {
    py::buffer_info ii = image.request();
    if (ii.ndim != 3) throw std::runtime_error("Only accepting single images");
    check_input(ii.shape[0], ii.shape[1], ii.shape[2]);
    const unet::Node &f = net_.nodes()[net_.final_layer()];
    run_forward(static_cast<const float *>(ii.ptr), 1, true);
    if (net_.loss() == LossFunction::kCrossEntropyLoss) {  // the class map: arg max over the logits, [1, H, W]
      py::array_t<float> cls({(py::ssize_t)1, (py::ssize_t)f.out_h, (py::ssize_t)f.out_w});
      net_.class_map_to_host(cls.mutable_data(), 1);
      return cls;
    }
    UNET_CALL(unet_threshold_f32(const_cast<float *>(f.act.f32()), 0.5f, f.out_elems(), net_.stream()));
    py::array_t<float> ret({(py::ssize_t)f.out_c, (py::ssize_t)f.out_h, (py::ssize_t)f.out_w});
    net_.copy_output_to_host(net_.final_layer(), ret.mutable_data(), 1);
    return ret;
  }

  // neural_net.h:25-44: Dice score of the thresholded prediction vs a uint8 mask.
  float test(farray image, py::array_t<uint8_t, py::array::c_style | py::array::forcecast> mask) {
    py::buffer_info ii = image.request(), mi = mask.request();
    if (ii.ndim != 3) throw std::runtime_error("Only accepting single images");
    check_input(ii.shape[0], ii.shape[1], ii.shape[2]);
    const unet::Node &f = net_.nodes()[net_.final_layer()];
    if (net_.loss() == LossFunction::kCrossEntropyLoss)
      throw std::runtime_error("test() returns a Dice score, which a class map does not have; use infer()");
    const py::ssize_t mh = mi.ndim >= 2 ? mi.shape[mi.ndim - 2] : -1, mw = mi.ndim >= 2 ? mi.shape[mi.ndim - 1] : -1;
    if (mh != f.out_h || mw != f.out_w || mi.size != (py::ssize_t)f.out_h * f.out_w)
      throw std::runtime_error("Mask dimensions do not match image dimensions");
    run_forward(static_cast<const float *>(ii.ptr), 1, true);
    UNET_CALL(unet_threshold_f32(const_cast<float *>(f.act.f32()), 0.5f, f.out_elems(), net_.stream()));
    std::vector<float> m((size_t)mi.size);
    const uint8_t *mp = static_cast<const uint8_t *>(mi.ptr);
    for (size_t i = 0; i < m.size(); ++i) m[i] = mp[i] ? 1.0f : 0.0f;
    UNET_CALL(unet_memcpy_h2d(net_.input_masks_ptr(), m.data(), sizeof(float) * m.size(), net_.stream()));
    float dice = 0.f;
    UNET_CALL(unet_dice_score_f32(net_.input_masks_ptr(), f.act.f32(), f.out_h, f.out_w, 1, net_.loss_values_ptr(),
                                  &dice, net_.stream()));
    return dice;
  }

  // ---- additive API -------------------------------------------------------------
  void set_seed(uint64_t seed) {
    if (net_.created()) net_.init_parameters(seed);
    else net_.set_seed(seed);
  }
  int64_t num_parameters() const { return net_.num_parameters(); }
  py::array_t<float> get_parameters() {
    py::array_t<float> a((py::ssize_t)net_.num_parameters());
    net_.copy_params_to_host(a.mutable_data());
    return a;
  }
  void set_parameters(farray p) {
    if (p.size() != net_.num_parameters()) throw std::runtime_error("set_parameters: wrong number of values");
    net_.copy_params_from_host(p.data());
  }
  py::array_t<float> get_gradients() {
    py::array_t<float> a((py::ssize_t)net_.num_parameters());
    net_.copy_grads_to_host(a.mutable_data());
    return a;
  }
  // [(layer index, offset, n_weights_and_biases, n_bn_params)] for layers with parameters
  std::vector<std::tuple<int, int64_t, int64_t, int64_t>> parameter_layout() const {
    std::vector<std::tuple<int, int64_t, int64_t, int64_t>> out;
    for (size_t i = 0; i < net_.nodes().size(); ++i) {
      const unet::Node &n = net_.nodes()[i];
      if (n.num_wb + n.num_bn > 0) out.emplace_back((int)i, n.param_off, n.num_wb, n.num_bn);
    }
    return out;
  }
  std::vector<std::vector<int>> output_shapes() const {
    std::vector<std::vector<int>> out;
    for (const unet::Node &n : net_.nodes()) out.push_back(n.output_shape);
    return out;
  }
  // images [N,C,H,W] -> final activations [N,Co,Ho,Wo] (no threshold)
  py::array_t<float> forward(farray images, bool inference) {
    py::buffer_info ii = images.request();
    if (ii.ndim != 4) throw std::runtime_error("images must have shape [N, C, H, W]");
    check_input(ii.shape[1], ii.shape[2], ii.shape[3]);
    const int batch = (int)ii.shape[0];
    run_forward(static_cast<const float *>(ii.ptr), batch, inference);
    return layer_output(net_.final_layer(), batch);
  }
  py::array_t<float> layer_output(int layer, int batch) {
    const unet::Node &n = net_.nodes().at(layer);
    py::array_t<float> ret({(py::ssize_t)batch, (py::ssize_t)n.out_c, (py::ssize_t)n.out_h, (py::ssize_t)n.out_w});
    net_.copy_output_to_host(layer, ret.mutable_data(), batch);
    return ret;
  }
  py::array_t<float> running_stats(int layer) {
    const unet::Node &n = net_.nodes().at(layer);
    py::array_t<float> a((py::ssize_t)n.num_bn);
    if (n.num_bn) net_.copy_running_stats_to_host(layer, a.mutable_data());
    return a;
  }
  void set_running_stats(int layer, farray stats) {
    const unet::Node &n = net_.nodes().at(layer);
    py::buffer_info bi = stats.request();
    if (bi.size != n.num_bn) throw std::runtime_error("set_running_stats: expected " + std::to_string(n.num_bn) + " values");
    if (n.num_bn) net_.copy_running_stats_from_host(layer, static_cast<const float *>(bi.ptr));
  }
  py::array_t<float> get_optimizer_state() {
    py::array_t<float> a((py::ssize_t)(2 * net_.num_parameters()));
    net_.copy_optimizer_state_to_host(a.mutable_data());
    return a;
  }
  void set_optimizer_state(farray state, int step_count) {
    py::buffer_info bi = state.request();
    if (bi.size != 2 * net_.num_parameters()) throw std::runtime_error("set_optimizer_state: wrong length");
    net_.copy_optimizer_state_from_host(static_cast<const float *>(bi.ptr), step_count);
  }
  int optimizer_steps() const { return net_.step_count(); }
  // Watch for NaN / Inf gradients on every step (also switched on by UNET_CHECK_NAN=1): train(), train_step()
  // raise once such a step has been seen; nonfinite_step() returns that step (1-based) or 0.
  void set_check_nan(bool on) { net_.set_check_nan(on); }
  int nonfinite_step() { return net_.nonfinite_step(); }
  // forward + loss + backward (+ Adam step when lr > 0 ... see do_step); returns the loss
  float train_step(farray images, farray masks, float learning_rate, bool do_step) {
    py::buffer_info ii = images.request(), mi = masks.request();
    if (ii.ndim != 4 || mi.ndim != 3) throw std::runtime_error("need images [N,C,H,W] and masks [N,H,W]");
    check_input(ii.shape[1], ii.shape[2], ii.shape[3]);
    check_mask(ii.shape[0], mi.shape[0], mi.shape[1], mi.shape[2]);
    const int batch = (int)ii.shape[0];
    net_.train_batch_host(static_cast<const float *>(ii.ptr), static_cast<const float *>(mi.ptr), batch,
                          learning_rate, do_step);
    return net_.last_loss(batch);
  }
  // ---- device-pointer entry points (data-parallel driver, benchmarks) ----------
  void reserve(int batch) { net_.ensure_batch(batch); }
  void set_stream(uintptr_t s) { net_.set_stream(reinterpret_cast<unet_stream_t>(s)); }
  uintptr_t parameters_ptr() const { return reinterpret_cast<uintptr_t>(net_.param_ptr()); }
  uintptr_t gradients_ptr() const { return reinterpret_cast<uintptr_t>(net_.grad_ptr()); }
  uintptr_t loss_values_ptr() const { return reinterpret_cast<uintptr_t>(net_.loss_values_ptr()); }
  void forward_backward_device(uintptr_t images, uintptr_t masks, int batch) {
    const float *im = reinterpret_cast<const float *>(images), *mk = reinterpret_cast<const float *>(masks);
    net_.forward_device(im, batch, false);
    net_.loss_forward_device(mk, batch);
    net_.backward_device(im, mk, batch);
  }
  void forward_device(uintptr_t images, int batch, bool inference) {
    net_.forward_device(reinterpret_cast<const float *>(images), batch, inference);
  }
  uintptr_t output_ptr() const {
    return reinterpret_cast<uintptr_t>(net_.nodes()[net_.final_layer()].act.f32());
  }
  void step(float learning_rate) { net_.step_device(learning_rate); }
  void synchronize() { net_.sync(); }
  void set_bucket_bytes(int64_t b) { net_.set_bucket_bytes(b); }
  // [(offset_floats, count_floats, event_handle)] in completion order
  std::vector<std::tuple<int64_t, int64_t, uintptr_t>> gradient_buckets() {
    std::vector<std::tuple<int64_t, int64_t, uintptr_t>> out;
    const auto &b = net_.buckets();
    for (size_t i = 0; i < b.size(); ++i)
      out.emplace_back(b[i].offset, b[i].count, reinterpret_cast<uintptr_t>(net_.bucket_event(i)));
    return out;
  }
  // data-parallel training on a communicator of the C ABI (handle from unet_comm_init_rank; 0 = none)
  void set_comm(uintptr_t comm) { net_.set_comm(reinterpret_cast<unet_comm_t>(comm)); }
  void broadcast_state(int root) { net_.broadcast_state(root); }
  // after writing through parameters_ptr(): packed weights and cached inference coefficients are stale
  void parameters_changed() { net_.parameters_changed(); }
  float last_epoch_loss() const { return last_epoch_loss_; }
  std::string compute_dtype() const { return net_.is_bf16() ? "bf16" : "f32"; }

 private:
  void check_input(int64_t C, int64_t H, int64_t W) {
    if (!net_.created()) throw std::runtime_error("create() must be called before using the network");
    if (C != net_.in_c() || H != net_.in_h() || W != net_.in_w()) {
      std::stringstream ss;
      ss << "image shape [" << C << ", " << H << ", " << W << "] does not match the network input [" << net_.in_c()
         << ", " << net_.in_h() << ", " << net_.in_w() << "]";
      throw std::runtime_error(ss.str());
    }
  }
  // Masks have the size of the network OUTPUT: the image size for the padded convolutions of main.py (the
  // reference's check, neural_net.h:137-140), smaller for graphs of unpadded convolutions.
  void check_mask(int64_t n_images, int64_t n_masks, int64_t h, int64_t w) {
    const unet::Node &f = net_.nodes()[net_.final_layer()];
    if (n_images != n_masks || h != f.out_h || w != f.out_w)
      throw std::runtime_error("Mask dimensions do not equal image dimensions");
  }
  void run_forward(const float *host_images, int batch, bool inference) {
    net_.ensure_batch(batch);
    UNET_CALL(unet_memcpy_h2d(net_.input_images_ptr(), host_images,
                              sizeof(float) * (size_t)batch * net_.in_c() * net_.in_h() * net_.in_w(), net_.stream()));
    net_.forward_device(net_.input_images_ptr(), batch, inference);
  }
  Net net_;
  float last_epoch_loss_ = 0.f;
};

// -------------------------------------------------------------- Convolution --
// Single-convolution helper (src/cuda_library/2Dconvolution/convolution_numpy.h:12-269).
class Convolution {
 public:
  // image [C_in,H,W], kernel [C_in,C_out,k,k], biases [C_out] -> flat fp32 array
  // of C_out*H_out*W_out values (the caller reshapes it).
  py::array_t<float> convolution2D(farray image, farray kernel, farray biases, int padding, int stride) {
    py::buffer_info ii = image.request(), ki = kernel.request(), bi = biases.request();
    if (ii.ndim != 3) throw std::runtime_error("Only accepting single images");
    if (ki.ndim != 4) {
      std::stringstream ss;
      ss << "Only accepting kernels with the format [C_in][C_out][H][W]. Found size of " << ki.ndim << std::endl;
      throw std::runtime_error(ss.str());
    }
    const int Ci = (int)ii.shape[0], H = (int)ii.shape[1], W = (int)ii.shape[2];
    if (Ci != ki.shape[0]) {
      std::stringstream ss;
      ss << "Kernel C_in must match input C_in: " << Ci << ", " << ki.shape[0] << std::endl;
      throw std::runtime_error(ss.str());
    }
    const int Co = (int)ki.shape[1], kh = (int)ki.shape[2], kw = (int)ki.shape[3];
    if (kh != kw || kh % 2 == 0) throw std::runtime_error("kernel must be square and odd.");
    if (bi.size != Co) throw std::runtime_error("biases must have one value per output channel");
    const int Ho = (H + 2 * padding - kh) / stride + 1, Wo = (W + 2 * padding - kw) / stride + 1;
    if (Ho < 1 || Wo < 1) throw std::runtime_error("convolution output would be empty");
    const size_t nw = (size_t)kh * kw * Ci * Co, nin = (size_t)Ci * H * W, nout = (size_t)Co * Ho * Wo;
    unet::DeviceBuffer dimg, dw, dout;
    dimg.ensure(sizeof(float) * nin);
    dw.ensure(sizeof(float) * (nw + Co));
    dout.ensure(sizeof(float) * nout);
    UNET_CALL(unet_memcpy_h2d(dimg.raw(), ii.ptr, sizeof(float) * nin, nullptr));
    UNET_CALL(unet_memcpy_h2d(dw.raw(), ki.ptr, sizeof(float) * nw, nullptr));
    UNET_CALL(unet_memcpy_h2d(dw.f32() + nw, bi.ptr, sizeof(float) * Co, nullptr));
    UNET_CALL(unet_convolve_f32(dw.f32(), dimg.f32(), dout.f32(), (int64_t)nout, 1, kh, Ci, Co, Ho, Wo, H, W, padding,
                                stride, nullptr));
    py::array_t<float> out((py::ssize_t)nout);
    UNET_CALL(unet_memcpy_d2h(out.mutable_data(), dout.raw(), sizeof(float) * nout, 1, nullptr));
    return out;
  }

  // 100 Adam steps (t = 1..100) fitting one convolution to target y with the
  // difference loss; returns weights then biases (convolution_numpy.h:26-191).
  py::array_t<float> train(farray input_image, farray y_image, farray weights, farray biases, int padding,
                           int stride, float learning_rate) {
    py::buffer_info ii = input_image.request(), yi = y_image.request(), wi = weights.request(), bi = biases.request();
    if (ii.ndim != 3 || yi.ndim != 3 || wi.ndim != 4)
      throw std::runtime_error("train: need input [C_in,H,W], y [C_out,H_out,W_out], weights [C_in,C_out,k,k]");
    const int Ci = (int)ii.shape[0], H = (int)ii.shape[1], W = (int)ii.shape[2];
    const int Co = (int)yi.shape[0], Ho = (int)yi.shape[1], Wo = (int)yi.shape[2];
    const int kh = (int)wi.shape[2], kw = (int)wi.shape[3];
    if (kh != kw || kh % 2 == 0) throw std::runtime_error("kernel must be square and odd.");
    if (Ci != wi.shape[0] || Co != wi.shape[1] || bi.size != Co)
      throw std::runtime_error("train: weights/biases do not match the input and target channel counts");
    if (Ho != (H + 2 * padding - kh) / stride + 1 || Wo != (W + 2 * padding - kw) / stride + 1)
      throw std::runtime_error("train: target size does not match the convolution output size");
    const size_t nw = (size_t)kh * kw * Ci * Co, nwb = nw + Co, nin = (size_t)Ci * H * W, nout = (size_t)Co * Ho * Wo;
    unet::DeviceBuffer dx, dy, dw, dout, dLdY, dLdW, adam;
    dx.ensure(sizeof(float) * nin);
    dy.ensure(sizeof(float) * nout);
    dw.ensure(sizeof(float) * nwb);
    dout.ensure(sizeof(float) * nout);
    dLdY.ensure(sizeof(float) * nout);
    dLdW.ensure(sizeof(float) * nwb);
    adam.ensure(sizeof(float) * 2 * nwb);
    UNET_CALL(unet_memcpy_h2d(dx.raw(), ii.ptr, sizeof(float) * nin, nullptr));
    UNET_CALL(unet_memcpy_h2d(dy.raw(), yi.ptr, sizeof(float) * nout, nullptr));
    UNET_CALL(unet_memcpy_h2d(dw.raw(), wi.ptr, sizeof(float) * nw, nullptr));
    UNET_CALL(unet_memcpy_h2d(dw.f32() + nw, bi.ptr, sizeof(float) * Co, nullptr));
    for (int t = 0; t < 100; ++t) {
      UNET_CALL(unet_convolve_f32(dw.f32(), dx.f32(), dout.f32(), (int64_t)nout, 1, kh, Ci, Co, Ho, Wo, H, W, padding,
                                  stride, nullptr));
      UNET_CALL(unet_difference_loss_backward_f32(dy.f32(), dout.f32(), dLdY.f32(), (int64_t)nout, nullptr));
      UNET_CALL(unet_convolve_backward_f32(dLdY.f32(), dx.f32(), dw.f32(), nullptr, dLdW.f32(), 1, kh, Ci, Co, Ho, Wo,
                                           H, W, padding, stride, nullptr));
      UNET_CALL(unet_update_weights_adam_f32(dw.f32(), dLdW.f32(), adam.f32(), learning_rate, (int64_t)nwb, t + 1,
                                             nullptr));
    }
    py::array_t<float> out((py::ssize_t)nwb);
    UNET_CALL(unet_memcpy_d2h(out.mutable_data(), dw.raw(), sizeof(float) * nwb, 1, nullptr));
    return out;
  }
};

}  // namespace

PYBIND11_MODULE(neural_net, m) {
  m.doc() = "MI355X (gfx950) implementation of the U-Net layer module";

  py::class_<NN>(m, "NeuralNet")
      .def(py::init<>())
      .def("create", &NN::create)
      .def("train", &NN::train)
      .def("infer", &NN::infer)
      .def("set_input_normalization", &NN::set_input_normalization)
      .def("train_u8", &NN::train_u8)
      .def("infer_u8", &NN::infer_u8)
      .def("test", &NN::test)
      // additive
      .def("set_seed", &NN::set_seed)
      .def("num_parameters", &NN::num_parameters)
      .def("get_parameters", &NN::get_parameters)
      .def("set_parameters", &NN::set_parameters)
      .def("get_gradients", &NN::get_gradients)
      .def("parameter_layout", &NN::parameter_layout)
      .def("output_shapes", &NN::output_shapes)
      .def("forward", &NN::forward, py::arg("images"), py::arg("inference") = false)
      .def("layer_output", &NN::layer_output)
      .def("running_stats", &NN::running_stats)
      .def("set_running_stats", &NN::set_running_stats)
      .def("get_optimizer_state", &NN::get_optimizer_state)
      .def("set_optimizer_state", &NN::set_optimizer_state)
      .def("optimizer_steps", &NN::optimizer_steps)
      .def("train_step", &NN::train_step, py::arg("images"), py::arg("masks"), py::arg("learning_rate"),
           py::arg("do_step") = true)
      .def("reserve", &NN::reserve)
      .def("set_stream", &NN::set_stream)
      .def("parameters_ptr", &NN::parameters_ptr)
      .def("gradients_ptr", &NN::gradients_ptr)
      .def("loss_values_ptr", &NN::loss_values_ptr)
      .def("output_ptr", &NN::output_ptr)
      .def("forward_backward_device", &NN::forward_backward_device)
      .def("forward_device", &NN::forward_device)
      .def("step", &NN::step)
      .def("synchronize", &NN::synchronize)
      .def("set_bucket_bytes", &NN::set_bucket_bytes)
      .def("gradient_buckets", &NN::gradient_buckets)
      .def("set_comm", &NN::set_comm)
      .def("broadcast_state", &NN::broadcast_state)
      .def("parameters_changed", &NN::parameters_changed)
      .def("last_epoch_loss", &NN::last_epoch_loss)
      .def("set_check_nan", &NN::set_check_nan)
      .def("nonfinite_step", &NN::nonfinite_step)
      .def("compute_dtype", &NN::compute_dtype);

  py::class_<Convolution>(m, "Convolution")
      .def(py::init<>())
      .def("convolution2D", &Convolution::convolution2D)
      .def("train", &Convolution::train);

  py::class_<UNetTest>(m, "UNetTest")
      .def(py::init<>())
      .def("testConvolution", &UNetTest::testConvolution)
      .def("testUpsample", &UNetTest::testUpsample)
      .def("testBatchNorm", &UNetTest::testBatchNorm)
      .def("testMaxPool", &UNetTest::testMaxPool);

  py::enum_<LossFunction>(m, "LossFunction")
      .value("DICE_LOSS", LossFunction::kDiceLoss)
      .value("DIFFERENCE_LOSS", LossFunction::kDifferenceLoss)
      .value("CROSS_ENTROPY_LOSS", LossFunction::kCrossEntropyLoss)
      .export_values();

  py::enum_<Layer_type>(m, "LayerType")
      .value("MAX_POOL_LAYER", Layer_type::MAX_POOL_LAYER)
      .value("FC_LAYER", Layer_type::FC_LAYER)
      .value("CONV_LAYER", Layer_type::CONV_LAYER)
      .value("ATTENTION_LAYER", Layer_type::ATTENTION_LAYER)
      .value("BN_LAYER", Layer_type::BN_LAYER)
      .value("UPSAMPLING_LAYER", Layer_type::UPSAMPLING_LAYER)
      .value("CONCAT_LAYER", Layer_type::CONCAT_LAYER)
      .export_values();

  py::enum_<Activation_type>(m, "ActivationType")
      .value("ReLU", Activation_type::ReLU)
      .value("SOFTMAX", Activation_type::SOFTMAX)
      .value("NONE", Activation_type::NONE)
      .value("TANH", Activation_type::TANH)
      .value("SIGMOID", Activation_type::SIGMOID)
      .value("LEAKY_ReLU", Activation_type::LEAKY_ReLU);

  py::class_<LayerDesc>(m, "LayerDesc")
      .def(py::init<const Layer_type &, const std::vector<int> &, const std::vector<int> &, const Activation_type &,
                    double>(),
           py::arg("type"), py::arg("shape"), py::arg("parents"), py::arg("activation") = NONE,
           py::arg("leaky_const") = 0.01)
      .def_readwrite("type", &LayerDesc::type)
      .def_readwrite("shape", &LayerDesc::descriptor)
      .def_readwrite("parents", &LayerDesc::parents)
      .def_readwrite("activation", &LayerDesc::activation)
      .def_readwrite("leaky_const", &LayerDesc::leaky_const);
}
