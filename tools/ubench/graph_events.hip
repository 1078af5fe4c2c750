// Can the training step be replayed as THREE graphs (main / skip / weight-gradient chains) launched on the engine's three streams, with the
// fork / join events between them captured as EXTERNAL event nodes (hipEventRecordWithFlags(hipEventRecordExternal) /
// hipStreamWaitEvent(hipEventWaitExternal))?    The step's dependencies run BOTH ways (main forks to the side chain,
// the side chain joins back), so it only works if a captured wait node waits for the record node OF THE SAME REPLAY even when the graph that
// holds the wait is launched before the graph that holds the record.  This probe measures exactly that on gfx950 / ROCm 7.2:
//   graph A (stream a): [spin 300 us] -> record(ev, external)
//   graph B (stream b): wait(ev, external) -> [stamp kernel]
// replayed three times in each launch order.  If B's stamp precedes the end of A's spin when B is launched FIRST, a wait node takes the event's
// state at the time it is enqueued (the previous replay's record, long complete): a join wait in the main graph would not wait for the side
// graph launched after it.
//   build:  hipcc --offload-arch=gfx950 -O2 tools/ubench/graph_events.hip -o tools/ubench/graph_events
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void spin_kernel(unsigned long long* stamps, int slot, int us) {
    const unsigned long long t0 = wall_clock64();
    stamps[2 * slot] = t0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) { __builtin_amdgcn_s_sleep(8); }
    stamps[2 * slot + 1] = wall_clock64();
}
__global__ void stamp_kernel(unsigned long long* stamps, int slot) { stamps[2 * slot] = wall_clock64(); stamps[2 * slot + 1] = wall_clock64(); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    unsigned long long* st; CK(hipMalloc(&st, 64 * sizeof(unsigned long long)));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipGraph_t ga, gb; hipGraphExec_t xa, xb;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, st, 0, 300);
    CK(hipEventRecordWithFlags(ev, a, hipEventRecordExternal));
    CK(hipStreamEndCapture(a, &ga));
    CK(hipStreamBeginCapture(b, hipStreamCaptureModeRelaxed));
    CK(hipStreamWaitEvent(b, ev, hipEventWaitExternal));
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, b, st, 1);
    CK(hipStreamEndCapture(b, &gb));
    CK(hipGraphInstantiate(&xa, ga, nullptr, nullptr, 0));
    CK(hipGraphInstantiate(&xb, gb, nullptr, nullptr, 0));
    size_t na = 0, nb = 0; CK(hipGraphGetNodes(ga, nullptr, &na)); CK(hipGraphGetNodes(gb, nullptr, &nb));
    printf("graph A: %zu nodes, graph B: %zu nodes (kernel + external event node each)\n", na, nb);
    for (int order = 0; order < 2; ++order)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(st, 0, 64 * sizeof(unsigned long long)));
            CK(hipDeviceSynchronize());
            if (order == 0) { CK(hipGraphLaunch(xa, a)); CK(hipGraphLaunch(xb, b)); }
            else { CK(hipGraphLaunch(xb, b)); CK(hipGraphLaunch(xa, a)); }
            CK(hipDeviceSynchronize());
            unsigned long long h[4]; CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
            const double spin_end = (h[1] - h[0]) * 0.01, stamp = ((double)h[2] - (double)h[0]) * 0.01;
            printf("launch order %s rep %d: A's spin ends at %.1f us, B's kernel behind the wait runs at %+.1f us -> %s\n", order == 0 ? "A, B" : "B, A", rep, spin_end, stamp,
                   stamp >= spin_end ? "waited for THIS replay's record" : "did NOT wait (saw an earlier record / none)");
        }
    return 0;
}
