// Micro-benchmark: read-modify-write of two float3 planes (796 MB total at 7680x4320) with
// (A) the strip pattern: lane i owns 48 contiguous bytes -> three float4 at stride 48 B
// (B) fully contiguous float4: lane i owns bytes [16 i, 16 i + 16) of each 1 KiB wave chunk
// (C) per-pixel 12-byte accesses (the straight kernel's pattern)
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k_strip(float4* p, float4* w, int strips_per_row, int rows, size_t pitch16)
{
    const int tx = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (tx >= strips_per_row || y >= rows) return;
    float4* a = p + (size_t)y * pitch16 + (size_t)tx * 3;
    float4* b = w + (size_t)y * pitch16 + (size_t)tx * 3;
    float4 a0 = a[0], a1 = a[1], a2 = a[2], b0 = b[0], b1 = b[1], b2 = b[2];
    a0.x += 1; a1.y += 1; a2.z += 1; b0.x += 1; b1.y += 1; b2.z += 1;
    a[0] = a0; a[1] = a1; a[2] = a2; b[0] = b0; b[1] = b1; b[2] = b2;
}
__global__ void __launch_bounds__(256) k_contig(float4* p, float4* w, int strips_per_row, int rows, size_t pitch16)
{
    // a wave covers the same 3 KiB per plane, but each instruction is a contiguous 1 KiB
    const int wave_tx0 = blockIdx.x * 64, lane = threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (wave_tx0 + lane >= strips_per_row || y >= rows) return;
    float4* a = p + (size_t)y * pitch16 + (size_t)wave_tx0 * 3;
    float4* b = w + (size_t)y * pitch16 + (size_t)wave_tx0 * 3;
    float4 a0 = a[lane], a1 = a[64 + lane], a2 = a[128 + lane], b0 = b[lane], b1 = b[64 + lane], b2 = b[128 + lane];
    a0.x += 1; a1.y += 1; a2.z += 1; b0.x += 1; b1.y += 1; b2.z += 1;
    a[lane] = a0; a[64 + lane] = a1; a[128 + lane] = a2; b[lane] = b0; b[64 + lane] = b1; b[128 + lane] = b2;
}
struct __attribute__((packed, aligned(4))) pix3 { float x, y, z; };
__global__ void __launch_bounds__(256) k_pix(pix3* p, pix3* w, int width, int rows, size_t pitch)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= width || y >= rows) return;
    pix3* a = (pix3*)((char*)p + (size_t)y * pitch) + x;
    pix3* b = (pix3*)((char*)w + (size_t)y * pitch) + x;
    pix3 u = *a, v = *b;
    u.x += 1; v.y += 1;
    *a = u; *b = v;
}

int main()
{
    const int W = 7680, H = 4320;
    const size_t bytes = (size_t)W * H * 12;
    void *p, *w;
    hipMalloc(&p, bytes); hipMalloc(&w, bytes);
    hipMemset(p, 0, bytes); hipMemset(w, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int strips = W / 4;
    const size_t pitch16 = (size_t)W * 12 / 16;
    for (int rep = 0; rep < 2; rep++)
        for (int v = 0; v < 3; v++) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) {
                if (v == 0) hipLaunchKernelGGL(k_strip, dim3((strips + 63) / 64, H / 4), dim3(64, 4), 0, 0, (float4*)p, (float4*)w, strips, H, pitch16);
                if (v == 1) hipLaunchKernelGGL(k_contig, dim3((strips + 63) / 64, H / 4), dim3(64, 4), 0, 0, (float4*)p, (float4*)w, strips, H, pitch16);
                if (v == 2) hipLaunchKernelGGL(k_pix, dim3(W / 64, H / 4), dim3(64, 4), 0, 0, (pix3*)p, (pix3*)w, W, H, (size_t)W * 12);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%s: %.3f ms per pass, %.2f TB/s (read+write of %.0f MB)\n", v == 0 ? "strip (48B/lane, 3 x float4 stride 48)" : v == 1 ? "contiguous float4 per instruction      " : "12-byte per lane (pix3)                ", ms, 4.0 * bytes / ms / 1e9, 2.0 * bytes / 1e6);
        }
    return 0;
}
