// Throughput of per-lane divergent gathers (one 128-B line per lane per step), the access pattern
// of the estimator's march step.  Variants differ in the loads issued per step.
//   0: 2 unaligned 8-B loads (o, o+25) + 1 byte load at +125      (current brick fetch)
//   1: 2 unaligned 8-B loads
//   2: 1 unaligned 8-B load
//   3: 2 aligned 8-B loads (o & ~7)
//   4: 1 aligned 16-B load
//   5: 2 unaligned 8-B loads + byte load, but all lanes of a wave in 8 lines (coherent rays)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V>
__global__ __launch_bounds__(256) void gather(const unsigned char *__restrict__ buf, unsigned log2_lines, unsigned steps,
                                              unsigned magic, unsigned long long *out)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    const unsigned mask = (1u << log2_lines) - 1u;
    unsigned long long acc = 0;
    for (unsigned i = 0; i < steps; i++) {
        s = s * 1664525u + 1013904223u + (unsigned)acc; // dependent on the loaded data
        unsigned line = (s >> 8) & mask;
        if (V == 5) {
            line = (line & ~7u & ~(63u << 3)) | ((threadIdx.x & 7u)) | (((blockIdx.x * 4u + (threadIdx.x >> 6)) * 2654435761u >> 12) & (mask & ~7u));
            line &= mask;
        }
        const unsigned o = (s >> 3) % 94u;
        const unsigned char *p = buf + ((size_t)line << 7);
        if (V == 0 || V == 5) {
            uint2 a = *(const uint2 *)(p + o);
            uint2 b = *(const uint2 *)(p + o + 25);
            unsigned m = p[125];
            acc += (a.x ^ a.y ^ b.x ^ b.y) + m;
        } else if (V == 1) {
            uint2 a = *(const uint2 *)(p + o);
            uint2 b = *(const uint2 *)(p + o + 25);
            acc += (a.x ^ a.y ^ b.x ^ b.y);
        } else if (V == 2) {
            uint2 a = *(const uint2 *)(p + o);
            acc += (a.x ^ a.y);
        } else if (V == 3) {
            uint2 a = *(const uint2 *)(p + (o & ~7u));
            uint2 b = *(const uint2 *)(p + ((o + 25) & ~7u));
            acc += (a.x ^ a.y ^ b.x ^ b.y);
        } else if (V == 4) {
            uint4 a = *(const uint4 *)(p + (o & ~15u));
            acc += (a.x ^ a.y ^ a.z ^ a.w);
        }
    }
    if (acc == magic) {
        atomicAdd(out, acc);
    }
}

template <int V>
static void run(const unsigned char *d, unsigned log2_lines, unsigned long long *dout, const char *name)
{
    const unsigned steps = 2000, blocks = 256 * 6;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(gather<V>, dim3(blocks), dim3(256), 0, 0, d, log2_lines, 200u, 0x9e3779b9u, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL(gather<V>, dim3(blocks), dim3(256), 0, 0, d, log2_lines, steps, 0x9e3779b9u, dout);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double fetches = (double)blocks * 256 * steps;
    printf("%-44s lines 2^%u (%4u MiB): %7.2f ms  %7.2f Gfetch/s  %6.1f cycles/wave-step/CU-slot\n", name, log2_lines,
           (1u << log2_lines) >> 13, ms, fetches / ms / 1e6, ms * 1e-3 * 2.4e9 / (steps * 24.0));
}

int main()
{
    for (unsigned log2_lines : { 15u, 21u, 23u }) { // 4 MiB (L2), 256 MiB, 1 GiB
        unsigned char *d;
        unsigned long long *dout;
        const size_t bytes = (size_t)128 << log2_lines;
        hipMalloc(&d, bytes + 256);
        hipMalloc(&dout, 8);
        hipMemset(d, 1, bytes + 256);
        hipMemset(dout, 0, 8);
        run<0>(d, log2_lines, dout, "2x unaligned 8B + byte (current)");
        run<1>(d, log2_lines, dout, "2x unaligned 8B");
        run<2>(d, log2_lines, dout, "1x unaligned 8B");
        run<3>(d, log2_lines, dout, "2x aligned 8B");
        run<4>(d, log2_lines, dout, "1x aligned 16B");
        run<5>(d, log2_lines, dout, "2x unaligned 8B + byte, 8 lines per wave");
        hipFree(d);
        hipFree(dout);
    }
    return 0;
}
