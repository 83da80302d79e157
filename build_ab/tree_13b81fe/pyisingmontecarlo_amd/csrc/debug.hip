// libisingmc.so: measurement hook -- the shader clock held while the sweep kernels run.
#include "internal.hpp"

// ------------------------------------------------------------------------------------------------
// measurement hook: the shader clock the chip holds WHILE the sweep kernels run (bench.py reports it next
// to the vector-ALU bound).  One wave on a side stream stamps s_memtime (shader cycles) and s_memrealtime
// (100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6) around a sleep loop of `probe_ms`, while `timesteps`
// sweeps run on the engine's stream.  The probe's exit condition is the constant-rate counter: every wave leaves.
// ------------------------------------------------------------------------------------------------
__global__ void clock_probe_kernel(unsigned long long *out, const unsigned long long realtime_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(100);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < realtime_ticks);
    out[0] = __builtin_amdgcn_s_memtime() - c0;
    out[1] = r1 - r0;
}

extern "C" int isingmc_debug_shader_clock(isingmc_states *s, size_t timesteps, double beta, double probe_ms, double *ghz_out)
{
    if (!s || !ghz_out) return fail(ISINGMC_ERR_INVALID, "NULL argument");
    if (!(probe_ms > 0.0) || probe_ms > 100.0) return fail(ISINGMC_ERR_INVALID, "probe_ms must be in (0, 100]");
    TRY(use_device(s->g->device));
    DeviceScratch scratch(s->stream);
    unsigned long long *d_out = nullptr, h_out[2] = {0, 0};
    TRY(scratch.alloc(&d_out, 2));
    struct Side { // a stream of its own: the replica lanes carry sweeps
        hipStream_t st = nullptr;
        ~Side() { if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); } }
    } side_owner;
    HIP_TRY(hipStreamCreateWithFlags(&side_owner.st, hipStreamNonBlocking));
    hipStream_t side = side_owner.st;
    // the sweeps first (they ramp the chip up), then the probe beside them; both are awaited
    int rc = run_steps(s, timesteps / 4, &beta, 0, nullptr, nullptr, /*sync=*/false);
    if (rc != ISINGMC_OK) return rc;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, side, d_out, (unsigned long long)(probe_ms * 1e5));
    HIP_TRY(hipGetLastError());
    rc = run_steps(s, timesteps - timesteps / 4, &beta, 0, nullptr, nullptr, /*sync=*/true);
    HIP_TRY(hipStreamSynchronize(side));
    if (rc != ISINGMC_OK) return rc;
    HIP_TRY(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    *ghz_out = h_out[1] ? double(h_out[0]) / double(h_out[1]) * 0.1 : 0.0; // cycles per 10 ns tick -> GHz
    return ISINGMC_OK;
}

