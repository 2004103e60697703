// fp32-compute mode of the UNet engine (csrc/unet_f32.hip): plain fp32 NCHW kernels, layer by layer.
#pragma once
#include "common.hpp"
#include <string>
#include <vector>

namespace bndm {

struct F32Model;
// names / values: the state dict (diffusers keys, PyTorch layouts) as loaded through bndm_unet_load_param
int f32_model_create(const bndm_unet_config &cfg, const std::vector<std::string> &names,
                     const std::vector<std::vector<float>> &values, F32Model **out);
void f32_model_destroy(F32Model *m);
// sample [B][Cin][R][R] (or [B][Cin/2][R][R] + extra of the same shape), timesteps [B], out [B][Cout][R][R]; device fp32
int f32_model_forward(F32Model *m, const float *sample, const float *extra, const float *timesteps, float *out, int B,
                      hipStream_t st);

}  // namespace bndm
