// What HBM delivers for the access shapes of the FFT-domain convolution kernels, with nothing else in the way:
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_patterns.hip -o /tmp/hbm_patterns && /tmp/hbm_patterns
// A persistent grid (W workgroups of 256 threads per CU) moves "rows" of `piece` contiguous bytes that lie `stride` bytes
// apart (a GEMM workgroup's window rows; a row transform's frequency pieces), `depth` 16-byte loads in flight per
// thread before the first store, in three modes: copy (read + write, what the GEMM kernels do), read only, write only.
// Reference points: piece = stride (a plain streaming copy), the packed form's GEMM (512 B / 295 424 B), the row form's
// GEMM (5120 B / 1 479 680 B), a row transform (256 B / 5120 B).  The tensors are 1.5 GB each (beyond the Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(256) void move_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t piece,
                                                    size_t stride, size_t pieces_per_task, size_t ntasks, size_t task_stride, size_t tpg,
                                                    size_t span, int mode, float *sink) {
    // task t = pieces_per_task pieces, `stride` apart, starting at (t / tpg) * span + (t % tpg) * task_stride; a workgroup
    // takes tasks round-robin
    const size_t per_piece = piece / 16;                    // float4 per piece
    const size_t n = pieces_per_task * per_piece;           // float4 per task
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const size_t start = (t / tpg) * span + (t % tpg) * task_stride;
        const unsigned char *s = src + start;
        unsigned char *d = dst + start;
        for (size_t i0 = threadIdx.x; i0 < n; i0 += 256 * DEPTH) {
            f4 v[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const size_t i = i0 + (size_t)u * 256;
                const size_t off = (i / per_piece) * stride + (i % per_piece) * 16;
                if (mode != 2) v[u] = i < n ? *reinterpret_cast<const f4 *>(s + off) : f4{0.f, 0.f, 0.f, 0.f};
                else v[u] = f4{1.f, 2.f, 3.f, (float)i};
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const size_t i = i0 + (size_t)u * 256;
                const size_t off = (i / per_piece) * stride + (i % per_piece) * 16;
                if (mode != 1) { if (i < n) *reinterpret_cast<f4 *>(d + off) = v[u]; }
                else acc += v[u];
            }
        }
    }
    if (mode == 1 && acc.x == 12345.f) sink[0] = acc.y;
}

static double run(int depth, const unsigned char *src, unsigned char *dst, size_t piece, size_t stride, size_t ppt, size_t ntasks,
                  size_t task_stride, size_t tpg, size_t span, int mode, int wg_per_cu, float *sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    auto launch = [&]() {
        if (depth == 4) move_kernel<4><<<grid, 256>>>(src, dst, piece, stride, ppt, ntasks, task_stride, tpg, span, mode, sink);
        else if (depth == 8) move_kernel<8><<<grid, 256>>>(src, dst, piece, stride, ppt, ntasks, task_stride, tpg, span, mode, sink);
        else move_kernel<16><<<grid, 256>>>(src, dst, piece, stride, ppt, ntasks, task_stride, tpg, span, mode, sink);
    };
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    const double bytes = (double)ntasks * ppt * piece * (mode == 0 ? 2 : 1);
    return bytes / (best * 1e-3) / 1e12;
}

int main() {
    const size_t total = (size_t)1536 << 20;                // 1.5 GB per tensor
    unsigned char *src, *dst; float *sink;
    hipMalloc(&src, total); hipMalloc(&dst, total); hipMalloc(&sink, 64);
    hipMemset(src, 1, total); hipMemset(dst, 0, total);
    struct Shape { const char *name; size_t piece, stride, ppt, task_stride, tpg; };
    // a task = what one workgroup pass moves: (GEMM) the window rows of one chunk for one frequency pair; (row) the pieces
    // of one image row / pair group.  tpg tasks share a group of `ppt x stride` bytes.
    const Shape shapes[] = {
        {"streaming copy, 64 KB tasks", 65536, 65536, 1, 65536, 1},
        {"packed GEMM: 64 window rows of 512 B, 295 424 B apart", 512, 295424, 64, 512, 577},
        {"row-form GEMM: 8 window blocks of 5120 B, 1 479 680 B apart", 5120, 1479680, 8, 5120, 289},
        {"row transform: 289 pieces of 256 B, 5120 B apart", 256, 5120, 289, 256, 20},
        {"packed row transform: 577 pieces of 128 B, 512 B apart", 128, 512, 577, 128, 4},
    };
    const char *modes[] = {"copy", "read", "write"};
    for (const Shape &sh : shapes) {
        const size_t span = sh.stride * sh.ppt;
        const size_t groups = total / span;
        const size_t ntasks = groups * sh.tpg;
        printf("%s  (%zu tasks, %.0f MB)\n", sh.name, ntasks, (double)ntasks * sh.ppt * sh.piece / 1e6);
        for (int mode = 0; mode < 3; ++mode)
            for (int wg = 2; wg <= 8; wg *= 2)
                for (int depth = 4; depth <= 16; depth *= 2)
                    printf("  %-5s %d WG/CU depth %2d: %.2f TB/s\n", modes[mode], wg, depth,
                           run(depth, src, dst, sh.piece, sh.stride, sh.ppt, ntasks, sh.task_stride, sh.tpg, span, mode, wg, sink));
        fflush(stdout);
    }
    return 0;
}
