// Slab reduction of the weight gradients, shared by its own launches (conv_wgrad.hip) and by the grouped weight-gradient launch, which
// carries the reduction of the PREVIOUS group's slabs as one more job (conv_wgrad_tile.hip).
#pragma once
#include "common.h"
#include "kernels.h"

// sum the split slabs and scatter into PyTorch layout  dst[n][c][tap]   (one job per conv layer)
// block / nblocks: this workgroup's place among the workgroups that share the job (256 threads each)
__device__ __forceinline__ void wgrad_reduce_body(const PaWgradReduceJob j, int block, int nblocks) {
    const int K = j.taps * j.Cin;
    const int total = j.real_cout * j.real_cin * j.taps;
    for (int e = block * 256 + (int)threadIdx.x; e < total + j.real_cout; e += nblocks * 256) {
        if (e < total) {
            // e enumerates the SOURCE order (n, tap, c) so that reads are coalesced
            const int n = e / (j.taps * j.real_cin);
            const int r = e - n * j.taps * j.real_cin;
            const int tap = r / j.real_cin, c = r - tap * j.real_cin;
            const float* src = j.part + (size_t)n * K + tap * j.Cin + c;
            // (16 loads in flight per thread: the split counts of the networks are multiples of 16 or small; the sum stays in split order)
            float s = 0.f;
            int sp = 0;
            for (; sp + 16 <= j.splits; sp += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u) * j.Cout * K];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += v[u];
            }
            if (sp < j.splits) {                       // the rest in one more batch (clamped, unconditional loads)
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u < j.splits ? sp + u : sp) * j.Cout * K];
#pragma unroll
                for (int u = 0; u < 16; ++u) if (sp + u < j.splits) s += v[u];
            }
            j.dst[((size_t)n * j.real_cin + c) * j.taps + tap] = s;
        } else if (j.dbdst) {
            const int n = e - total;
            // (16 loads in flight, the sum in split order: the plain loop compiled to one load per wait -- splits x ~0.9 us, 60 us for the
            // 16 threads of an output layer's bias while the rest of the launch was long done)
            float s = 0.f;
            if (j.dbpart) {
                const float* src = j.dbpart + n;
                int sp = 0;
                for (; sp + 16 <= j.splits; sp += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u) * j.Cout];
#pragma unroll
                    for (int u = 0; u < 16; ++u) s += v[u];
                }
                if (sp < j.splits) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u < j.splits ? sp + u : sp) * j.Cout];
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (sp + u < j.splits) s += v[u];
                }
            }
            j.dbdst[n] = s;
        }
    }
}

