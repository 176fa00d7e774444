// Cost of a kernel boundary inside a replayed hipGraph on this chip: N dependent launches of an (almost) empty kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(float* p, int work) {
    float v = p[threadIdx.x];
    for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}
static void run(int blocks, int threads, int work, int n) {
    float* d; (void)hipMalloc(&d, 4096);
    hipStream_t s; (void)hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) tiny<<<blocks, threads, 0, s>>>(d, work);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    (void)hipGraphLaunch(ge, s);
    (void)hipEventRecord(e1, s);
    (void)hipStreamSynchronize(s);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d x %4d, %5d flops/thread: %7.2f us per launch (graph of %d)\n", blocks, threads, work, ms * 1e3 / n, n);
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g); (void)hipStreamDestroy(s); (void)hipFree(d);
}
int main() {
    run(1, 64, 0, 400); run(256, 256, 0, 400); run(512, 256, 0, 400); run(2048, 256, 0, 400); run(512, 512, 0, 400);
    run(512, 256, 1000, 400); run(512, 256, 4000, 400);
    return 0;
}
