// Descriptor types of the Python-facing API.  Names and enumerator order follow
// the reference (src/neural_net_implementation/layer.h:21-44) because they are
// part of the module's surface (src/bindings/net.cpp:49-87).
#pragma once
#include <vector>

enum Layer_type {
  CONV_LAYER,
  BN_LAYER,
  MAX_POOL_LAYER,
  ATTENTION_LAYER,
  FC_LAYER,
  CONCAT_LAYER,
  UPSAMPLING_LAYER,
};

enum Activation_type { ReLU, SOFTMAX, NONE, TANH, SIGMOID, LEAKY_ReLU };

enum class LossFunction {
  kDiceLoss,
  kDifferenceLoss,
  // not in the reference (neural_net.h:18 has the two above): softmax cross-entropy over the channels of the
  // final layer, masks holding class indices; the final layer's SOFTMAX activation (layer.h:31, which the
  // reference declares and runs as ReLU) is then the identity and the softmax lives in the loss
  kCrossEntropyLoss,
};

// descriptor field orders (layer.h:426-432, 461-465, 580-584):
//   CONV_LAYER        [H_in, W_in, C_in, C_out, kernel, padding, stride]
//   MAX_POOL_LAYER    [C, H, W, kernel, stride]
//   UPSAMPLING_LAYER  [H_in, W_in, C_in, C_out, scale]
//   CONCAT_LAYER      []   (shape from the two parents; first parent = skip)
//   ATTENTION_LAYER   []   (parents = [x (skip), g (gating, half the size of x)]; layer.h:521-571)
struct LayerDesc {
  Layer_type type;
  std::vector<int> descriptor;
  std::vector<int> parents;
  Activation_type activation;
  double leaky_const;
  LayerDesc(const Layer_type &t, const std::vector<int> &d, const std::vector<int> &p,
            const Activation_type &a, double lc)
      : type(t), descriptor(d), parents(p), activation(a), leaky_const(lc) {}
};
