// Which streams run beside the default stream?  A long kernel on the null stream, a tiny one on a candidate stream
// queued right after it: the candidate overlaps when its kernel ends long before the long one does.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void busy(unsigned long long* out, unsigned long long cycles)
    {
    extern __shared__ unsigned long long lds[];
    lds[threadIdx.x] = cycles; // (36 KB of LDS per workgroup: four workgroups per CU, half its wave slots stay free)
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < cycles)
        t = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        out[0] = t;
    }
__global__ void tiny(unsigned long long* out)
    {
    out[0] = wall_clock64();
    }

int main()
    {
    unsigned long long* d;
    CK(hipMalloc(&d, 1024));
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("priority range: least %d greatest %d\n", least, greatest);
    std::vector<hipStream_t> s;
    for (int i = 0; i < 6; i++)
        {
        hipStream_t x;
        CK(hipStreamCreateWithPriority(&x, hipStreamNonBlocking, 0));
        s.push_back(x);
        }
    hipStream_t hp;
    CK(hipStreamCreateWithPriority(&hp, hipStreamNonBlocking, greatest));
    s.push_back(hp);
    hipEvent_t b0, b1, t1, ready;
    CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1)); CK(hipEventCreate(&t1));
    CK(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    // wall_clock64 ticks at 100 MHz: 30000 ticks = 300 us; grid: 256 CUs x 8 workgroups of 256 threads, twice over
    for (int rep = 0; rep < 3; rep++)
        for (size_t i = 0; i < s.size(); i++)
            {
            CK(hipDeviceSynchronize());
            // a small kernel first on the null stream (the role of the integration pass), then the event, then the long one
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, nullptr, d + 8);
            CK(hipEventRecord(ready, nullptr));
            CK(hipEventRecord(b0, nullptr));
            hipLaunchKernelGGL(busy, dim3(256 * 4 * 4), dim3(256), 36 * 1024, nullptr, d, 7500ull);
            CK(hipEventRecord(b1, nullptr));
            CK(hipStreamWaitEvent(s[i], ready, 0));
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i], d + 16);
            CK(hipEventRecord(t1, s[i]));
            CK(hipDeviceSynchronize());
            float big = 0, small_end = 0;
            CK(hipEventElapsedTime(&big, b0, b1));
            CK(hipEventElapsedTime(&small_end, b0, t1));
            printf("rep %d stream %zu%s: long kernel %.1f us, tiny kernel of the other stream done at %.1f us -> %s\n", rep, i,
                   i + 1 == s.size() ? " (high priority)" : "", big * 1e3, small_end * 1e3, small_end < 0.2f * big ? "OVERLAPS" : "waits");
            }
    return 0;
    }
