// Prints which XCD (XCC_ID hardware register) each workgroup of a grid lands on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
}
int main()
{
    const int n = 64;
    unsigned *d;
    hipMalloc(&d, 2 * n * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) {
        printf("block %2d xcc %u hwid %08x\n", i, h[2 * i], h[2 * i + 1]);
    }
    return 0;
}
