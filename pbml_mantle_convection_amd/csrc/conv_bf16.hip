// bf16 MFMA convolution path — placeholder until the MFMA kernels land (returns MC_EUNSUPPORTED
// loudly; there is no silent fallback to another precision).
#include "conv_common.h"
int mc_bf16_tile(const mc_conv_desc* d, int* th, int* tw) { (void)d; *th = 16; *tw = 16; return MC_EUNSUPPORTED; }
size_t mc_bf16_bank_bytes(const ConvGeom& g, int dgrad) { (void)g; (void)dgrad; return 0; }
int mc_bf16_pack(const ConvGeom& g, const float* w, int dgrad, void* packed, hipStream_t s) { (void)g; (void)w; (void)dgrad; (void)packed; (void)s; return MC_EUNSUPPORTED; }
int mc_conv2d_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* bank, const float* bias, void* y0, void* y1, float* part, hipStream_t s) { (void)g; (void)x0; (void)x1; (void)bank; (void)bias; (void)y0; (void)y1; (void)part; (void)s; return MC_EUNSUPPORTED; }
int mc_wgrad_bf16(const ConvGeom& g, const void* x0, const void* x1, const void* dy, void* part, hipStream_t s) { (void)g; (void)x0; (void)x1; (void)dy; (void)part; (void)s; return MC_EUNSUPPORTED; }
const char* mc_bf16_kernel_name(const ConvGeom& g) { (void)g; return "unsupported"; }
