// split_gemm_test.cpp - accuracy and rate of the split-bf16 tile GEMM (cmdgen_amd/csrc/cmdgen_split.h) on its own.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/split_gemm_test tools/split_gemm_test.cpp && tools/split_gemm_test
//
// (1) accuracy: C = A W^T for A [M,256] (SiLU of normals), W [256,256] uniform +-1/16, against fp64 on the host,
//     next to the error of an fp32 fmaf chain (= v_mfma_f32_32x32x2_f32, bitwise) on the same data;
// (2) rate: a chain of G GEMMs per MT-row tile (each GEMM's SiLU'd output is the next one's input, as in the node
//     kernel), G weight matrices streamed from L2, one workgroup per tile, two workgroups per CU.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include "../cmdgen_amd/csrc/cmdgen_split.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int H = 256;
struct WList { const void* w[8]; };

__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

template <int MT>
__global__ __launch_bounds__(256, 2) void k_chain_rs(int M, const float* __restrict__ A, WList Ws, int G,
                                                     float* __restrict__ out, int raw_out, int epi) {
    constexpr int LDAF = H + 4;
    __shared__ __attribute__((aligned(16))) float buf[MT * LDAF];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * MT;
    SCarry carry;
    SFragPtr f = sfrag_ptr(Ws.w[0], H / 16, 0, wave);
    split_prefetch(f, carry);
    {
        const int c4 = tid % 64, rsub = tid / 64;
#pragma unroll 4
        for (int pass = 0; pass < MT / 4; ++pass) {
            const int r = pass * 4 + rsub;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < M) v = reinterpret_cast<const float4*>(A + (size_t)(row0 + r) * H)[c4];
            *reinterpret_cast<float4*>(buf + r * LDAF + 4 * c4) = v;
        }
    }
    __syncthreads();
    sf32x16 acc[MT / 32][2];
    for (int g = 0; g < G; ++g) {
        const SFragPtr fn = sfrag_ptr(Ws.w[(g + 1) % G], H / 16, 0, wave);
#pragma unroll
        for (int m = 0; m < MT / 32; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        tile_gemm_rsplit<MT, H / 16>(buf, LDAF, f, fn, acc, carry);
        f = fn;
        __syncthreads();
        if (g + 1 < G && epi) {
#pragma unroll
            for (int m = 0; m < MT / 32; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = wave * 64 + n * 32 + (lane & 31);
                        buf[row * LDAF + col] = silu_f(acc[m][n][r]);
                    }
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < MT / 32; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = wave * 64 + n * 32 + (lane & 31);
                if (row0 + row < M) out[(size_t)(row0 + row) * H + col] = raw_out ? acc[m][n][r] : silu_f(acc[m][n][r]);
            }
}

static unsigned short bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static float bf16_f(unsigned short b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

static std::vector<unsigned short> pack_split(const std::vector<float>& W, int out, int in) {
    const int NT = out / 32, KB = in / 16;
    std::vector<unsigned short> p((size_t)NT * KB * 3 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const float w = W[(size_t)(32 * nt + (lane & 31)) * in + 16 * kb + 8 * (lane >> 5) + j];
                    const unsigned short h0 = bf16_rne(w); const float r1 = w - bf16_f(h0);
                    const unsigned short h1 = bf16_rne(r1); const float r2 = r1 - bf16_f(h1);
                    const unsigned short h2 = bf16_rne(r2);
                    const size_t base = (((size_t)nt * KB + kb) * 3) * 64 * 8;
                    p[base + (0 * 64 + lane) * 8 + j] = h0; p[base + (1 * 64 + lane) * 8 + j] = h1; p[base + (2 * 64 + lane) * 8 + j] = h2;
                }
    return p;
}

template <int MT>
static void run_rs(int M, int G, const float* dA, const WList& Ws, float* dOut, const char* tag, int reps, int epi = 1) {
    const int grid = (M + MT - 1) / MT;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_chain_rs<MT>), dim3(grid), dim3(256), 0, 0, M, dA, Ws, G, dOut, 0, epi);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_chain_rs<MT>), dim3(grid), dim3(256), 0, 0, M, dA, Ws, G, dOut, 0, epi);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, flop = 2.0 * M * H * (double)H * G;
    printf("%-28s M=%6d G=%d grid=%5d  %8.2f us  %7.1f TF/s (fp32-equivalent)  weights streamed %.1f TB/s\n", tag, M, G, grid, us,
           flop / us * 1e-6, (double)grid * G * H * H * 6 / us * 1e-6);
}

int main() {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 2.f);
    std::uniform_real_distribution<float> ud(-1.f / 16, 1.f / 16);
    const int Mbig = 256 * 64 * 3, G = 7;
    std::vector<float> A((size_t)Mbig * H);
    for (auto& v : A) { const float x = nd(rng); v = x / (1.f + std::exp(-x)); }
    std::vector<std::vector<float>> W(G, std::vector<float>((size_t)H * H));
    for (auto& w : W) for (auto& v : w) v = ud(rng);
    float* dA; CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    WList Ws{};
    for (int g = 0; g < G; ++g) {
        auto p = pack_split(W[g], H, H);
        void* d; CHECK(hipMalloc(&d, p.size() * 2)); CHECK(hipMemcpy(d, p.data(), p.size() * 2, hipMemcpyHostToDevice));
        Ws.w[g] = d;
    }
    float* dOut; CHECK(hipMalloc(&dOut, A.size() * 4));

    // ---- accuracy: one GEMM, raw output ----
    const int Mv = 512;
    for (int mt : {64, 32}) {
        CHECK(hipMemset(dOut, 0, (size_t)Mv * H * 4));
        if (mt == 64) hipLaunchKernelGGL((k_chain_rs<64>), dim3(Mv / 64), dim3(256), 0, 0, Mv, dA, Ws, 1, dOut, 1, 1);
        else hipLaunchKernelGGL((k_chain_rs<32>), dim3(Mv / 32), dim3(256), 0, 0, Mv, dA, Ws, 1, dOut, 1, 1);
        CHECK(hipDeviceSynchronize());
        std::vector<float> out((size_t)Mv * H); CHECK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
        double emax = 0, e2 = 0, cmax = 0, c2 = 0, relmax = 0;
        for (int i = 0; i < Mv; ++i)
            for (int o = 0; o < H; ++o) {
                double ref = 0, den = 0; float chain = 0.f;
                for (int k = 0; k < H; ++k) {
                    const float a = A[(size_t)i * H + k], w = W[0][(size_t)o * H + k];
                    ref += (double)a * w; den += std::fabs((double)a * w); chain = fmaf(a, w, chain);
                }
                const double e = std::fabs(out[(size_t)i * H + o] - ref), c = std::fabs(chain - ref);
                emax = std::max(emax, e); e2 += e * e; cmax = std::max(cmax, c); c2 += c * c; relmax = std::max(relmax, e / den);
            }
        printf("accuracy MT=%d: split-bf16 max|err| %.3e rms %.3e (max / sum|ab| %.3e)   fp32 fmaf chain max %.3e rms %.3e\n", mt, emax,
               std::sqrt(e2 / (Mv * H)), relmax, cmax, std::sqrt(c2 / (Mv * H)));
    }
    // ---- rate ----
    for (int M : {3776, 15104, Mbig}) {
        run_rs<64>(M, G, dA, Ws, dOut, "MT=64", 20);
        run_rs<64>(M, G, dA, Ws, dOut, "MT=64 GEMM only", 20, 0);
        run_rs<32>(M, G, dA, Ws, dOut, "MT=32", 20);
        run_rs<32>(M, G, dA, Ws, dOut, "MT=32 GEMM only", 20, 0);
    }
    printf("done\n");
    return 0;
}
