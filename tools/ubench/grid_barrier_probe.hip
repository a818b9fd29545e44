// grid_barrier_probe.hip -- round-5 probe behind "split the temporal predictor's feed-forward of ONE sequence over 8-32 workgroups with a
// device-scope reduction per layer" (DESIGN.md section 9): what does one barrier between G workgroups cost on an MI355X, where
// the workgroups sit on different XCDs (eight L2s that are not coherent with each other: an agent-scope release is an L2 write-back, an
// acquire an L2 invalidate)?
//
// The kernel does what such a feed-forward would do per layer: every workgroup writes a 16 x 48 float partial result, releases, arrives on a
// counter, spins until all G have arrived (agent-scope acquire loads), then reads all G partials and sums them -- N times in a row.  Reported:
// microseconds per round, and the sum check.  A spin gives up after 2^24 polls (the kernel then reports a failure instead of hanging).
// G workgroups of 512 threads with no LDS to speak of are co-resident on an idle MI355X for every G probed here (<= 64 of 256 CUs).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/grid_barrier_probe tools/ubench/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 512, PART = 16 * 48;

__global__ __launch_bounds__(NT) void probe(float* partials /* [2][G][PART] */, unsigned* counter, float* out /* [G] */, int* failed, int n_rounds, int G)
{
    const int g = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < n_rounds; ++r) {
        float* mine = partials + ((size_t)(r & 1) * G + g) * PART;
        for (int i = tid; i < PART; i += NT) mine[i] = (float)(g + 1) + 1e-3f * (float)(r & 7);
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); // (release: this workgroup's partial reaches memory)
            const unsigned want = (unsigned)G * (unsigned)(r + 1);
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > (1u << 24)) { *failed = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __atomic_thread_fence(__ATOMIC_ACQUIRE); // (every thread: the others' partials are read from memory, not from a stale line)
        const float* all = partials + (size_t)(r & 1) * G * PART;
        for (int i = tid; i < PART; i += NT) {
            float s = 0.f;
            for (int k = 0; k < G; ++k) s += __builtin_nontemporal_load(all + (size_t)k * PART + i);
            acc += s;
        }
    }
    // (block-wide sum of acc, thread 0 stores it: the check that every round saw every partial)
    __shared__ float red[NT];
    red[tid] = acc;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) out[g] = red[0];
}

int main(int argc, char** argv)
{
    const int n_rounds = argc > 1 ? atoi(argv[1]) : 2000;
    for (int G : {1, 2, 4, 8, 16, 32, 64}) {
        float *partials, *out;
        unsigned* counter;
        int* failed;
        hipMalloc(&partials, sizeof(float) * 2 * G * PART);
        hipMalloc(&out, sizeof(float) * G);
        hipMalloc(&counter, sizeof(unsigned));
        hipMalloc(&failed, sizeof(int));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) { // (the last repetition is reported: the first ones warm the clock up)
            hipMemset(counter, 0, sizeof(unsigned));
            hipMemset(failed, 0, sizeof(int));
            hipEventRecord(e0);
            hipLaunchKernelGGL(probe, dim3(G), dim3(NT), 0, 0, partials, counter, out, failed, n_rounds, G);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        std::vector<float> h(G);
        int hf = 0;
        hipMemcpy(h.data(), out, sizeof(float) * G, hipMemcpyDeviceToHost);
        hipMemcpy(&hf, failed, sizeof(int), hipMemcpyDeviceToHost);
        // expected per workgroup: sum over rounds of PART * (G (G + 1) / 2 + G * 1e-3 * (r & 7))
        double want = 0;
        for (int r = 0; r < n_rounds; ++r) want += (double)PART * (G * (G + 1) / 2.0 + G * 1e-3 * (r & 7));
        double worst = 0;
        for (int g = 0; g < G; ++g) worst = std::max(worst, std::abs((double)h[g] - want) / want);
        printf("G = %2d workgroups: %7.3f us per round (write 3 KB, release, arrive, spin, acquire, read %d x 3 KB)   spin gave up: %s   worst relative sum error %.1e\n",
               G, ms * 1e3 / n_rounds, G, hf ? "YES" : "no", worst);
        hipFree(partials); hipFree(out); hipFree(counter); hipFree(failed);
    }
    return 0;
}
