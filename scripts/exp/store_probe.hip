// Store/load-pattern probe: the marching kernels' memory pattern without any arithmetic.
// Layout as in the library: slot[(x*NF + f)*ny + y], NF = 5.  A block owns W columns and walks CH rows of x;
// per row every lane stores (or loads) NF values.  Variants: bytes per lane (8 / 16), block width, alignment,
// cache-policy bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NF = 5;

// MODE 0: store 8 B/lane; 1: store 16 B/lane (lane handles 2 adjacent columns); 2: load 8 B/lane; 3: load 16 B/lane
template <int MODE, int AUX>
__global__ void probe(double* buf, int nx, int ny, int W, int ch, double* sink) {
    const int t = threadIdx.x;
    const int cols_per_lane = (MODE & 1) ? 2 : 1;
    const int y = (int)blockIdx.x * W + t * cols_per_lane;
    const bool on = t * cols_per_lane < W && y < ny;
    const int xs = (int)blockIdx.y * ch, xe = xs + ch < nx ? xs + ch : nx;
    const unsigned pbytes = (unsigned)((long)NF * ny * 8), fstride = (unsigned)ny * 8u;
    const unsigned voff = on ? (unsigned)y * 8u : 0xFFFFFFF0u;
    double acc = 0.0;
    for (int x = xs; x < xe; ++x) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(buf + (long)x * NF * ny, 0, pbytes, 0x00020000);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, (double)(x + f)), r, voff, f * fstride, AUX);
            if (MODE == 1) { v4i v = {x, f, t, 1}; __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, f * fstride, AUX); }
            if (MODE == 2) acc += __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, f * fstride, AUX));
            if (MODE == 3) { v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, f * fstride, AUX); acc += v.x + v.z; }
        }
    }
    if (MODE >= 2 && acc == 12345.678) sink[0] = acc;
}

template <int MODE, int AUX>
static void run(const char* name, double* buf, double* sink, int nx, int ny, int NT, int W, int ch) {
    dim3 grid((ny + W - 1) / W, (nx + ch - 1) / ch), block(NT);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<MODE, AUX>), grid, block, 0, 0, buf, nx, ny, W, ch, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<MODE, AUX>), grid, block, 0, 0, buf, nx, ny, W, ch, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize()); CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-34s NT %4d W %4d chunk %4d : %.3f ms  %.2f TB/s\n", name, NT, W, ch, ms, (double)nx * ny * NF * 8 / ms / 1e9);
}

int main() {
    const int nx = 4096, ny = 4096;
    double* buf; double* sink;
    CHECK(hipMalloc(&buf, (size_t)nx * ny * NF * 8)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, (size_t)nx * ny * NF * 8));
    run<0, 0>("store 8B/lane", buf, sink, nx, ny, 128, 126, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 128, 112, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 128, 128, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 256, 256, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 512, 512, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 1024, 1024, 32);
    run<0, 2>("store 8B/lane aligned nt", buf, sink, nx, ny, 128, 128, 32);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 128, 128, 128);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 128, 128, 8);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 128, 128, 1);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 256, 256, 1);
    run<0, 0>("store 8B/lane aligned", buf, sink, nx, ny, 1024, 1024, 1);
    run<1, 0>("store 16B/lane aligned", buf, sink, nx, ny, 64, 128, 32);
    run<1, 0>("store 16B/lane aligned", buf, sink, nx, ny, 128, 256, 32);
    run<1, 0>("store 16B/lane aligned", buf, sink, nx, ny, 256, 512, 32);
    run<1, 2>("store 16B/lane aligned nt", buf, sink, nx, ny, 128, 256, 32);
    run<1, 0>("store 16B/lane aligned", buf, sink, nx, ny, 256, 512, 1);
    run<2, 0>("load 8B/lane", buf, sink, nx, ny, 128, 126, 32);
    run<2, 0>("load 8B/lane aligned", buf, sink, nx, ny, 128, 128, 32);
    run<2, 0>("load 8B/lane aligned", buf, sink, nx, ny, 256, 256, 32);
    run<2, 0>("load 8B/lane aligned", buf, sink, nx, ny, 256, 256, 1);
    run<3, 0>("load 16B/lane aligned", buf, sink, nx, ny, 128, 256, 32);
    run<3, 0>("load 16B/lane aligned", buf, sink, nx, ny, 256, 512, 32);
    return 0;
}
