#include "../../deepestscatter_amd/csrc/ct_device.hpp"
using namespace ct;
extern "C" {
__global__ void k_newdir(const float* cdf, const uint16_t* guide, float* io, uint32_t* seeds) {
    const int i = threadIdx.x; uint32_t s = seeds[i]; f3 d = mk3(io[3*i], io[3*i+1], io[3*i+2]);
    d = new_direction(cdf, guide, s, d); io[3*i]=d.x; io[3*i+1]=d.y; io[3*i+2]=d.z; seeds[i]=s; }
__global__ void k_costheta(const float* cdf, const uint16_t* guide, float* io, uint32_t* seeds) {
    const int i = threadIdx.x; io[i] = sample_cos_theta(cdf, guide, seeds[i]); }
__global__ void k_sincos(float* io) { const int i = threadIdx.x; float s, c; ct_sincosf(io[i], &s, &c); io[i] = s; io[i+64] = c; }
__global__ void k_log(float* io) { const int i = threadIdx.x; io[i] = logf_above_one(io[i]); }
__global__ void k_exp(float* io) { const int i = threadIdx.x; io[i] = expf_inrange(io[i]); }
__global__ void k_div(float* io) { const int i = threadIdx.x; io[i] = io[i] / io[i+64]; }
__global__ void k_rcp(float* io) { const int i = threadIdx.x; io[i] = 1.0f / io[i]; }
__global__ void k_sqrt(float* io) { const int i = threadIdx.x; io[i] = sqrtf(io[i]); }
__global__ void k_rcpm(float* io) { const int i = threadIdx.x; io[i] = rcp_moderate(io[i]); }
__global__ void k_sqrtm(float* io) { const int i = threadIdx.x; io[i] = sqrt_moderate(io[i]); }
__global__ void k_tea(uint32_t* io) { const int i = threadIdx.x; io[i] = tea4(io[i], io[i+64]); }
__global__ void k_filter(DevScene sc, float* io, uint2* cells) { const int i = threadIdx.x; io[i] = filter_at(sc, cells[i], mk3(io[i], io[i+64], io[i+128])); }
__global__ void k_fetchm(DevScene sc, float* io, uint2* cells) { const int i = threadIdx.x; uint32_t m; cells[i] = fetch_cell_m<false>(sc, mk3(io[i], io[i+64], io[i+128]), m); io[i] = (float)m; }
__global__ void k_inbox(DevScene sc, float* io) { const int i = threadIdx.x; io[i] = in_box(sc, mk3(io[i], io[i+64], io[i+128])) ? 1.f : 0.f; }
__global__ void k_empty(float* io) { const int i = threadIdx.x; io[i] = io[i+64]; }
}
