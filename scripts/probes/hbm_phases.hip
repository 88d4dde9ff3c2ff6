// How much of the streaming rate survives a kernel's phase structure: the packed GEMM's access shape (64 window rows of
// 512 B, 295 424 B apart: scripts/probes/hbm_patterns.hip moves 5.6 TB/s through it as a plain copy) with a compute
// phase of `spin` cycles and a workgroup barrier between a task's loads and its stores, software-pipelined one task
// ahead (the loads of task t + 1 are issued before the compute phase of task t), W workgroups of 256 threads per CU.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_phases.hip -o /tmp/hbm_phases && /tmp/hbm_phases
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool SPREAD, bool CONTIG>
__global__ __launch_bounds__(256) void phase_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t piece, size_t stride,
                                                     size_t ntasks, size_t task_stride, size_t tpg, size_t span, int spin, int nparts) {
    __shared__ float sh[256];
    const size_t per_piece = piece / 16;
    auto addr = [&](size_t t, int u) -> size_t {
        const size_t start = (t / tpg) * span + (t % tpg) * task_stride;
        const size_t i = threadIdx.x + (size_t)u * 256;
        return start + (i / per_piece) * stride + (i % per_piece) * 16;
    };
    f4 cur[DEPTH], nxt[DEPTH];
    // CONTIG: the task order of a GEMM that keeps one frequency pair's weights in registers -- a workgroup walks a contiguous
    // range of (pair, chunk) tasks, chunk fastest: at any moment the workgroups of the chip are spread over all chunks and
    // pairs.  Else round-robin with the pair fastest: the workgroups of the chip read neighbouring rows of the same windows.
    const size_t groups = ntasks / tpg;
    const size_t per = (ntasks + gridDim.x - 1) / gridDim.x;
    const size_t q0 = CONTIG ? blockIdx.x * per : blockIdx.x, q1 = CONTIG ? (q0 + per < ntasks ? q0 + per : ntasks) : ntasks;
    const size_t qstep = CONTIG ? 1 : gridDim.x;
    auto task_of = [&](size_t q) -> size_t { return CONTIG ? (q % groups) * tpg + q / groups : q; };
    size_t q = q0;
    if (q < q1)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) nxt[u] = *reinterpret_cast<const f4 *>(src + addr(task_of(q), u));
    for (; q < q1; q += qstep) {
        const size_t t = task_of(q);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) cur[u] = nxt[u];
        sh[threadIdx.x] = cur[0].x;                          // (a use: the loads must have landed)
        __syncthreads();
        const bool has_next = q + qstep < q1;
        const size_t tn = has_next ? task_of(q + qstep) : t;
        if (!SPREAD && has_next)
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) nxt[u] = *reinterpret_cast<const f4 *>(src + addr(tn, u));
        // compute phase in `nparts` pieces; SPREAD: a share of the next task's loads and of this task's stores around each piece
        for (int part = 0; part < nparts; ++part) {
            if (SPREAD && has_next)
#pragma unroll
                for (int u = 0; u < DEPTH; ++u)
                    if (u * nparts / DEPTH == part) nxt[u] = *reinterpret_cast<const f4 *>(src + addr(tn, u));
            const long long t0 = clock64();
            while (clock64() - t0 < spin / nparts) { }
            if (SPREAD)
#pragma unroll
                for (int u = 0; u < DEPTH; ++u)
                    if (u * nparts / DEPTH == part) *reinterpret_cast<f4 *>(dst + addr(t, u)) = cur[u] + sh[(threadIdx.x + 1) & 255];
        }
        if (!SPREAD)
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) *reinterpret_cast<f4 *>(dst + addr(t, u)) = cur[u] + sh[(threadIdx.x + 1) & 255];
        __syncthreads();
    }
}

int main() {
    const size_t total = (size_t)1536 << 20;
    unsigned char *src, *dst;
    (void)hipMalloc(&src, total); (void)hipMalloc(&dst, total);
    (void)hipMemset(src, 1, total); (void)hipMemset(dst, 0, total);
    const size_t piece = 512, stride = 295424, ppt = 64, task_stride = 512, tpg = 577;
    const size_t span = stride * ppt, groups = total / span, ntasks = groups * tpg;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("packed-GEMM shape, 32 KB in + 32 KB out per task (depth 8 x 16 B x 256 threads), %zu tasks\n", ntasks);
    for (int spread = 0; spread < 3; ++spread)
        for (int wg = 2; wg <= 8; wg *= 2)
            for (int spin = 0; spin <= 8000; spin = spin ? spin * 2 : 1000) {
                auto launch = [&]() {
                    if (spread == 1) phase_kernel<8, true, false><<<256 * wg, 256>>>(src, dst, piece, stride, ntasks, task_stride, tpg, span, spin, 4);
                    else if (spread == 0) phase_kernel<8, false, false><<<256 * wg, 256>>>(src, dst, piece, stride, ntasks, task_stride, tpg, span, spin, 4);
                    else phase_kernel<8, false, true><<<256 * wg, 256>>>(src, dst, piece, stride, ntasks, task_stride, tpg, span, spin, 4);
                };
                launch(); (void)hipDeviceSynchronize();
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                printf("  %s %d WG/CU, compute phase %5d cycles per task: %.2f TB/s (%.2f us per task and workgroup)\n", spread == 1 ? "spread " : (spread == 0 ? "bursts " : "contig "),
                       wg, spin, (double)ntasks * ppt * piece * 2 / (best * 1e-3) / 1e12, best * 1e3 * 256 * wg / ntasks);
            }
    return 0;
}
