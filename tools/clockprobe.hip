// clockprobe.hip -- a one-wave kernel that samples the shader clock while other kernels run (tools/clocktrace.py).
// s_memtime counts shader cycles, wall_clock64 the constant 100 MHz reference: consecutive samples give the clock held between
// them.  Launched first, on its own stream, the probe keeps one wave slot of one CU for its whole life (16 registers, no LDS) and
// sleeps between samples.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/_build/libclockprobe.so tools/clockprobe.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void clockprobe_kernel(unsigned long long* buf, int n, int naps) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        buf[2 * i] = __builtin_amdgcn_s_memtime();
        buf[2 * i + 1] = wall_clock64();
        __builtin_nontemporal_store(buf[2 * i + 1], &buf[2 * i + 1]);
        for (int s = 0; s < naps; ++s) __builtin_amdgcn_s_sleep(127);       // 127 x 64 cycles
    }
}

extern "C" int clockprobe_launch(unsigned long long* buf, int n, int naps, void* stream) {
    hipLaunchKernelGGL(clockprobe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), buf, n, naps);
    return (int)hipGetLastError();
}
