// Trainable SmallEncoder (reference: serl_launcher/vision/small_encoders.py:9-55, selected at
// agents/continuous/drq.py:137-153): x/255 -> 4 x (Conv 3x3 stride 2 VALID + bias -> ReLU), features 32/64/128/256
// -> mean over (H, W); the Dense(256)+LayerNorm+tanh bottleneck that follows is the encoder head the agent already has.
#pragma once
#include "internal.h"

namespace serl {

constexpr int kSmallLayers = 4;
constexpr int kSmallFeat[kSmallLayers + 1] = {3, 32, 64, 128, 256};

struct SmallDims {
  int H, W;
  int h[kSmallLayers + 1], w[kSmallLayers + 1];   // [0] = input, [l+1] = output of conv l
};
SmallDims small_dims(int H, int W);
// floats of one camera's conv stack in the parameter arena: per layer [9*cin + 1][cout] = kernel (HWIO) then bias
long small_conv_params();
long small_conv_offset(int layer);   // offset of layer's kernel inside that block

struct SmallWorkspace {
  SmallDims d{};
  int max_images = 0;             // images per pass (all cameras together)
  float* col[kSmallLayers]{};     // (unused since round 4: layer 0 reads the u8 frames directly)
  int* tab[kSmallLayers]{};       // layers 1..3: offset of every im2col row's patch in the layer's NHWC input (implicit GEMM)
  bool tab_ready = false;
  float* act[kSmallLayers]{};     // ReLU outputs [rows_l][cout_l]
  float* dact = nullptr;          // gradient wrt a layer's output (ping)
  float* dact2 = nullptr;         // (pong)
  float* dcol = nullptr;          // gradient wrt an im2col matrix
  float* slabs = nullptr;         // split-K partials of the weight-gradient GEMMs
  long slabs_cap = 0;
  // layer 0 runs directly on the u8 frames (no im2col matrix): the backward pass re-reads the frames of the last forward pass
  const uint8_t* last_frames = nullptr;
  long last_frame_cam_stride = 0;
  size_t bytes = 0;
};
size_t small_workspace_bytes(int max_images, int H, int W);
int small_workspace_bind(SmallWorkspace& ws, void* mem, int max_images, int H, int W);

// frames: u8, image i of camera c at frames + (c * frame_cam_stride + i) * H*W*3 (device); P + cam*cam_stride + conv_off
// = that camera's conv parameters.  pooled: [n_cam][pooled_cam_stride / 256 rows][256] -- rows [0, n) of every camera
// block are written.
int small_forward(SmallWorkspace& ws, const float* P, long conv_off, long cam_stride, const uint8_t* frames,
                  long frame_cam_stride, int n_cam, int n, float* pooled, long pooled_cam_stride, hipStream_t stream);
// backward of the LAST small_forward (its im2col matrices and activations are still in the workspace):
// dpooled [n_cam][dp_cam_stride/256][256] -> parameter gradients into G + cam*cam_stride + conv_off (same layout as P)
int small_backward(SmallWorkspace& ws, const float* P, long conv_off, long cam_stride, int n_cam, int n, const float* dpooled,
                   long dp_cam_stride, float* G, hipStream_t stream);

}  // namespace serl
