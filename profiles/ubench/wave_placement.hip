// wave_placement.hip -- where the dispatcher puts the waves of a workgroup.
//
// The wave-specialised kernels (gr4j.hip gr4j_pipe_kernel, cemaneige.hip) give
// the waves of a workgroup different amounts of work, so it matters which SIMD
// of the CU each wave lands on.  This prints, for grids of G workgroups of W
// waves (all co-resident: every wave spins until the host-visible deadline),
// the histogram "waves per SIMD" and, per wave index within the workgroup,
// the histogram of SIMD ids.
//
//   hipcc --offload-arch=gfx950 -O2 -o wave_placement wave_placement.hip
//   ./wave_placement
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <vector>

// HW_REG_HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
__global__ void where(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the wave resident for a while so that the whole grid coexists
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main()
{
    const int cfgs[][2] = {{1954, 1}, {1954, 2}, {1954, 3}, {1954, 7},
                           {977, 4},  {512, 2},  {15625, 2}, {1954, 6}};
    for (auto &c : cfgs) {
        const int G = c[0], W = c[1];
        unsigned *d;
        hipMalloc(&d, sizeof(unsigned) * 2 * G * W);
        hipMemset(d, 0xff, sizeof(unsigned) * 2 * G * W);
        where<<<G, 64 * W>>>(d, 400000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(2 * G * W);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        hipFree(d);
        std::map<unsigned, int> per_simd;          // key: xcc, se, sh, cu, simd
        std::vector<std::vector<int>> by_index(W, std::vector<int>(4, 0));
        int same_cu = 0;
        for (int g = 0; g < G; ++g) {
            unsigned cu0 = 0;
            for (int w = 0; w < W; ++w) {
                const unsigned hw = h[2 * (g * W + w)], xcc = h[2 * (g * W + w) + 1];
                const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xff;
                const unsigned key = ((xcc & 0xf) << 12) | (cu << 4) | simd;
                per_simd[key]++;
                by_index[w][simd]++;
                if (w == 0) cu0 = (xcc << 8) | cu;
                else if (((xcc << 8) | cu) == cu0) same_cu++;
            }
        }
        std::map<int, int> hist;
        for (auto &kv : per_simd) hist[kv.second]++;
        printf("grid %d x %d waves: %zu SIMDs used; waves per SIMD histogram:",
               G, W, per_simd.size());
        for (auto &kv : hist) printf("  %d waves: %d SIMDs", kv.first, kv.second);
        printf("\n");
        for (int w = 0; w < W; ++w)
            printf("   wave %d of its workgroup -> SIMD0..3: %d %d %d %d\n", w,
                   by_index[w][0], by_index[w][1], by_index[w][2],
                   by_index[w][3]);
    }
    return 0;
}
