// energy.hip -- what the headline kernel's ingredients cost at the socket.
//
// The HBV-Edu headline (1M sets x 10,957 days, qsim written) runs at the
// socket's 1,400 W cap with the shader clock pulled down to ~1.8 GHz: the
// sweep is bound by power, and the next instruction to remove should be chosen
// by watts, not by count.  This program holds the GPU in a steady state made
// of the kernel's ingredients, one at a time on top of each other, so that a
// driver (profiles/ubench/energy.py) can read socket power and shader clock
// from hwmon for each:
//
//   mode 0  v_fma_f64 only: 32 per trip (8 independent chains x 4), four
//           waves per SIMD on every CU -- the kernel's ~35 fp64 instructions
//           a set-day;
//   mode 1  + the scalar mix of a day: one 64-byte scalar load burst (the day
//           record) and 12 scalar ALU instructions per trip;
//   mode 2  + one 8-byte `nt sc1` buffer store per lane and trip (512 B per
//           wave: a row segment of qsim), rows of the [T][N] layout;
//   mode 3  + two LDS table reads per trip (the power's tables: 0.8 per day
//           in the kernel, two on the days that take the power);
//   mode 4  mode 2's stores with NO arithmetic (the store stream alone);
//   mode 5  mode 0 with 24 instead of 32 FMAs per trip plus mode 2's stores
//           (what removing a quarter of the arithmetic buys at the cap);
//   mode 6  the stores alone, TWO adjacent columns per lane: one 16-byte
//           store per lane and trip, 1 KiB per wave and row (2,048 waves);
//   mode 7  two sets per lane: 64 FMAs + scalar mix + one 16-byte store per
//           lane and trip (2,048 waves: the same sets as mode 2).
//
//   modes 10..15  mode 0 with another instruction in the FMA's place:
//           v_add_f64, v_mul_f64, v_max_f64, v_cndmask_b32 (VOP3), v_add_u32,
//           v_rcp_f64 (8 per trip instead of 32: quarter rate) -- what each
//           kind costs at the socket.
//
//   hipcc --offload-arch=gfx950 -O2 -o energy energy.hip
//   ./energy <mode> <seconds>     -> prints trips/s per wave and GB/s stored
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define WAVES_PER_SIMD 4
#define TRIPS 4096

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void soak(double *out, const double *rec,
                                           long ld, int trips, double b,
                                           double c)
{
    __shared__ double tab[1024];
    for (int j = threadIdx.x; j < 1024; j += 64) tab[j] = 1.0 + j * 1e-9;
    __syncthreads();
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 0.001 * (threadIdx.x + k);
    constexpr bool WIDE = MODE == 6 || MODE == 7;
    const long first = (long)blockIdx.x * (WIDE ? 128 : 64);
    double *row = out + first;
    const int lane_off = threadIdx.x * (WIDE ? 16 : 8);
    typedef const double __attribute__((address_space(4))) *cp_t;
    cp_t rp = (cp_t)rec;
    unsigned sx = blockIdx.x;
    const unsigned long long sx64 = 0x5555555555555555ull ^ blockIdx.x;
    constexpr bool STORES = MODE >= 2 && MODE < 10;
    constexpr bool SCALAR = MODE == 1 || MODE == 2 || MODE == 3 || MODE == 5 ||
                            MODE == 7;
    constexpr int FMAS = (MODE == 4 || MODE == 6 || MODE >= 10) ? 0
                         : (MODE == 5 ? 24 : (MODE == 7 ? 64 : 32));
    for (int t = 0; t < trips; ++t) {
        double r0 = 0, r1 = 0;
        if (SCALAR) {
            // the day record: one scalar load burst, used below
            asm volatile("" : "+s"(rp));
            r0 = rp[0]; r1 = rp[7];
            rp += 8;
            if ((t & 63) == 63) rp -= 512;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                asm volatile("s_add_u32 %0, %0, 0x9e37\n\ts_xor_b32 %0, %0, 0x5bd1"
                             : "+s"(sx) : : "scc");
            }
        }
        if constexpr (MODE >= 10) {
            int *ai = reinterpret_cast<int *>(a);
#pragma unroll
            for (int r = 0; r < (MODE == 15 ? 1 : 4); ++r) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (MODE == 10)
                        asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                    else if (MODE == 11)
                        asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                    else if (MODE == 12)
                        asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                    else if (MODE == 13)
                        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2"
                                     : "+v"(ai[2 * k]) : "v"(ai[2 * k + 1]), "s"(sx64));
                    else if (MODE == 14)
                        asm volatile("v_add_u32 %0, %0, %1" : "+v"(ai[2 * k]) : "v"(ai[2 * k + 1]));
                    else
                        asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));
                }
            }
        }
#pragma unroll
        for (int r = 0; r < FMAS / 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
        }
        if (SCALAR) {
            // (the record's values enter the arithmetic: the loads are real)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "s"(r0));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[1]) : "v"(b), "s"(r1));
        }
        if (MODE == 3) {
            const int j = (threadIdx.x * 13 + t) & 1023;
            a[2] += tab[j];
            a[3] += tab[(j + 517) & 1023];
        }
        if (STORES && !WIDE) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)row, (short)0, 512, 0x00020000);
            v2i d;
            d.x = __double2loint(a[t & 7]);
            d.y = __double2hiint(a[t & 7]);
            __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane_off, 0, 18);
            row += ld;
        }
        if (STORES && WIDE) {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void *)row, (short)0, 1024, 0x00020000);
            v4i d;
            d.x = __double2loint(a[t & 7]);
            d.y = __double2hiint(a[t & 7]);
            d.z = __double2loint(a[(t + 1) & 7]);
            d.w = __double2hiint(a[(t + 1) & 7]);
            __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane_off, 0, 18);
            row += ld;
        }
    }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    if (s == 12345.678 && sx == 77) out[0] = s;
}

int main(int argc, char **argv)
{
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return 1;
    const bool wide = mode == 6 || mode == 7;
    const int waves = pr.multiProcessorCount * 4 * WAVES_PER_SIMD / (wide ? 2 : 1);
    const long ld = (long)waves * (wide ? 128 : 64);
    double *out = nullptr, *rec = nullptr;
    const size_t bytes = (size_t)ld * 8 * (TRIPS + 1);
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("hipMalloc\n"); return 1; }
    (void)hipMalloc(&rec, 8 * 8 * (TRIPS + 600));
    (void)hipMemset(rec, 0, 8 * 8 * (TRIPS + 600));
    auto launch = [&]() {
        switch (mode) {
        case 0: soak<0><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 1: soak<1><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 2: soak<2><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 3: soak<3><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 4: soak<4><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 5: soak<5><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 6: soak<6><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 7: soak<7><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 10: soak<10><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 11: soak<11><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 12: soak<12><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 13: soak<13><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        case 14: soak<14><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        default: soak<15><<<waves, 64>>>(out, rec, ld, TRIPS, 0.999999, 1e-7); break;
        }
    };
    launch();
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int k = 0; k < 4; ++k) launch();
        (void)hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    const double trips_per_s = (double)launches * TRIPS / el;       // per wave
    const bool stores = mode >= 2 && mode < 10;
    printf("mode %d waves %d seconds %.2f trips_per_wave_per_s %.4e "
           "stored_GBps %.1f wave_trips_per_s %.4e\n", mode, waves, el,
           trips_per_s, stores ? trips_per_s * waves * (wide ? 1024 : 512) / 1e9 : 0.0,
           trips_per_s * waves);
    return 0;
}
