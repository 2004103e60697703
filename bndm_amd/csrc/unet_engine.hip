// placeholder until the engine lands (next commit)
#include "common.hpp"
using namespace bndm;
#define NOTYET(name) set_error(name ": UNet engine not built yet"); return BNDM_E_STATE
extern "C" int bndm_unet_create(bndm_unet **, const bndm_unet_config *) { NOTYET("bndm_unet_create"); }
extern "C" void bndm_unet_destroy(bndm_unet *) {}
extern "C" int bndm_unet_num_params(const bndm_unet *) { return 0; }
extern "C" int bndm_unet_param_info(const bndm_unet *, int, char *, size_t, int64_t *) { NOTYET("bndm_unet_param_info"); }
extern "C" int bndm_unet_load_param(bndm_unet *, const char *, const float *, int64_t) { NOTYET("bndm_unet_load_param"); }
extern "C" int bndm_unet_finalize(bndm_unet *) { NOTYET("bndm_unet_finalize"); }
extern "C" int bndm_unet_forward(bndm_unet *, const float *, const float *, float *, int, void *) { NOTYET("bndm_unet_forward"); }
extern "C" int bndm_unet_sample_iadb(bndm_unet *, float *, const float *, int, int, int, const float *, const float *, const float *, const uint8_t *, float *, void *) { NOTYET("bndm_unet_sample_iadb"); }
extern "C" int bndm_unet_sample_ddim(bndm_unet *, float *, int, int, const float *, float, void *) { NOTYET("bndm_unet_sample_ddim"); }
extern "C" int bndm_unet_profile(bndm_unet *, const float *, const float *, float *, int, int, bndm_unet_profile_t *, void *) { NOTYET("bndm_unet_profile"); }
