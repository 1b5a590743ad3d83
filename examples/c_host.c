/* c_host.c — the drop-in boundary used from plain C: no Python, no torch, only include/tip_hip.h + the HIP runtime.
 *
 * What a C/C++ host (or any FFI: cgo, JNI, N-API) does to run TF_RNN_Past_State.forward
 * (/root/reference/simple_transformer_with_state.py:60-102) on an MI355X:
 *   tip_create -> tip_tensor_info (the 56 state-dict tensors, same order as the reference's state_dict()) -> tip_pack_weights
 *   -> upload + tip_attach_packed -> tip_workspace_bytes -> tip_forward on a HIP stream.
 * Weights and inputs are a counter-based hash of (tensor index, element index), so that tests/test_c_host_gpu.py can rebuild the same
 * numbers in numpy, load them into the Python module and compare the two outputs bit for bit.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host.c -o c_host \
 *       -L transformer-inertial-poser_amd/csrc -ltip_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/transformer-inertial-poser_amd/csrc
 *   ./c_host y.bin [B] [T]         # writes B*T*131 float32, prints a checksum
 *   ./c_host y.bin B 40 reuse F    # F frames of B lock-stepped streams through tip_forward_reuse (a frame's in_linear
 *                                  # and layer-0 Q / K / V rows computed once, kept in a ring for the 40 windows it appears in):
 *                                  # windows grow 1 .. 40, then slide; writes F*B*131 float32 (row T-1 of every call)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tip_hip.h"

static uint32_t hash32(uint32_t z) {   /* lowbias32 */
    z ^= z >> 16; z *= 0x7FEB352Du; z ^= z >> 15; z *= 0x846CA68Bu; z ^= z >> 16;
    return z;
}
/* uniform in [-1, 1): element `i` of stream `stream` */
static float unit(uint32_t stream, uint32_t i) {
    const uint32_t h = hash32(i * 0x9E3779B1u + hash32(stream + 0x1234567u));
    return (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

#define CHECK_TIP(x) do { int s_ = (x); if (s_ < 0) { fprintf(stderr, "%s: %s\n", #x, tip_strerror(s_)); return 2; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "y.bin";
    const int B = argc > 2 ? atoi(argv[2]) : 3, T = argc > 3 ? atoi(argv[3]) : 40;
    tip_config cfg = {72, 131, 512, 1024, 256, 16, 4, 1, 1, 40};   /* the paper configuration (train_model.py defaults) */
    tip_handle* h = NULL;
    CHECK_TIP(tip_create(&cfg, &h));
    const int n = tip_num_tensors(h);
    if (n <= 0) return 2;

    /* the state-dict tensors, filled tensor by tensor: matrices ~ U(-1,1) / sqrt(fan_in), LayerNorm weights 1 + 0.1 u, other vectors 0.05 u */
    const float** tensors = (const float**)calloc((size_t)n, sizeof(float*));
    for (int i = 0; i < n; ++i) {
        const char* name; int rows, cols;
        CHECK_TIP(tip_tensor_info(h, i, &name, &rows, &cols));
        const size_t count = (size_t)rows * (size_t)(cols > 0 ? cols : 1);
        float* t = (float*)malloc(count * sizeof(float));
        const int is_norm_w = strstr(name, "norm") && strstr(name, "weight");
        float scale = 0.05f, shift = 0.0f;
        if (cols > 0) { scale = 1.0f; for (int k = 1; k * k < cols; ++k) scale = 1.0f / (float)(k + 1); }   /* ~ 1 / ceil(sqrt(cols)) */
        if (is_norm_w) { scale = 0.1f; shift = 1.0f; }
        for (size_t e = 0; e < count; ++e) t[e] = shift + scale * unit((uint32_t)i, (uint32_t)e);
        tensors[i] = t;
    }
    size_t packed_bytes = 0;
    CHECK_TIP(tip_packed_bytes(h, &packed_bytes));
    void* packed_host = malloc(packed_bytes);
    CHECK_TIP(tip_pack_weights(h, tensors, n, packed_host, packed_bytes));
    void* packed_dev = NULL;
    CHECK_HIP(hipMalloc(&packed_dev, packed_bytes));
    CHECK_HIP(hipMemcpy(packed_dev, packed_host, packed_bytes, hipMemcpyHostToDevice));
    CHECK_TIP(tip_attach_packed(h, packed_dev, packed_bytes));

    if (argc > 5 && !strcmp(argv[4], "reuse")) {
        /* ---- sliding windows with exact reuse.  Frame f of stream b has ONE input row (x_imu 90 + x_s 131 values, a hash of (b, f, column));
         *      call f sees frames max(0, f - 39) .. f.  The ring belongs to the caller, like every other buffer. ---- */
        const int F = atoi(argv[5]);
        size_t ring_bytes = 0, ws_bytes = 0;
        CHECK_TIP(tip_reuse_cache_bytes(h, B, &ring_bytes));
        CHECK_TIP(tip_workspace_bytes(h, B, 40, &ws_bytes));     /* (sizes grow with T: the full window's serves the growing ones) */
        void *ring = NULL, *ws = NULL;
        float *xi_d = NULL, *xs_d = NULL, *y_d = NULL;
        CHECK_HIP(hipMalloc(&ring, ring_bytes));
        CHECK_HIP(hipMalloc(&ws, ws_bytes));
        CHECK_HIP(hipMalloc((void**)&xi_d, (size_t)B * 40 * 90 * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&xs_d, (size_t)B * 40 * 131 * sizeof(float)));
        CHECK_HIP(hipMalloc((void**)&y_d, (size_t)B * 131 * sizeof(float)));
        float* xi = (float*)malloc((size_t)B * 40 * 90 * sizeof(float));
        float* xs = (float*)malloc((size_t)B * 40 * 131 * sizeof(float));
        float* y = (float*)malloc((size_t)B * 131 * sizeof(float));
        hipStream_t stream;
        CHECK_HIP(hipStreamCreate(&stream));
        CHECK_TIP(tip_reuse_reset(ring, ring_bytes, stream));
        FILE* fo = fopen(out_path, "wb");
        if (!fo) return 4;
        double sum = 0.0;
        for (int f = 0; f < F; ++f) {
            const int Tf = f + 1 < 40 ? f + 1 : 40, f0 = f + 1 - Tf;
            for (int b = 0; b < B; ++b)
                for (int t = 0; t < Tf; ++t) {
                    for (int c = 0; c < 90; ++c) xi[((size_t)b * Tf + t) * 90 + c] = unit(2000u + (uint32_t)b, (uint32_t)((f0 + t) * 90 + c));
                    for (int c = 0; c < 131; ++c) xs[((size_t)b * Tf + t) * 131 + c] = 0.5f * unit(3000u + (uint32_t)b, (uint32_t)((f0 + t) * 131 + c));
                }
            CHECK_HIP(hipMemcpy(xi_d, xi, (size_t)B * Tf * 90 * sizeof(float), hipMemcpyHostToDevice));
            CHECK_HIP(hipMemcpy(xs_d, xs, (size_t)B * Tf * 131 * sizeof(float), hipMemcpyHostToDevice));
            CHECK_TIP(tip_forward_reuse(h, xi_d, xs_d, y_d, B, Tf, TIP_FWD_LAST_ROW_ONLY, ring, ring_bytes, f, NULL, ws, ws_bytes, stream));
            CHECK_HIP(hipStreamSynchronize(stream));
            CHECK_HIP(hipMemcpy(y, y_d, (size_t)B * 131 * sizeof(float), hipMemcpyDeviceToHost));
            if (fwrite(y, sizeof(float), (size_t)B * 131, fo) != (size_t)B * 131) return 4;
            for (int e = 0; e < B * 131; ++e) sum += (double)y[e];
        }
        fclose(fo);
        CHECK_TIP(tip_check(h, 0));
        uint64_t nf = 0;
        CHECK_TIP(tip_forward_count(h, &nf));
        printf("c_host reuse: B=%d frames=%d sum=%.9g forwards=%llu\n", B, F, sum, (unsigned long long)nf);
        tip_destroy(h);
        return 0;
    }
    /* windows: x_imu [B,T,90] (72 IMU readings + the 18 acc-sum columns), x_s [B,T,131] */
    const size_t ni = (size_t)B * T * 90, ns = (size_t)B * T * 131;
    float* xi = (float*)malloc(ni * sizeof(float));
    float* xs = (float*)malloc(ns * sizeof(float));
    for (size_t e = 0; e < ni; ++e) xi[e] = unit(1000u, (uint32_t)e);
    for (size_t e = 0; e < ns; ++e) xs[e] = 0.5f * unit(1001u, (uint32_t)e);
    float *xi_d = NULL, *xs_d = NULL, *y_d = NULL;
    void* ws = NULL;
    size_t ws_bytes = 0;
    CHECK_TIP(tip_workspace_bytes(h, B, T, &ws_bytes));
    CHECK_HIP(hipMalloc((void**)&xi_d, ni * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&xs_d, ns * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&y_d, ns * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_HIP(hipMemcpy(xi_d, xi, ni * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(xs_d, xs, ns * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_TIP(tip_forward(h, xi_d, xs_d, y_d, B, T, 0, NULL, 1.0f, ws, ws_bytes, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_TIP(tip_check(h, 0));   /* a lost inter-workgroup hand-off is an error, not a number */
    float* y = (float*)malloc(ns * sizeof(float));
    CHECK_HIP(hipMemcpy(y, y_d, ns * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (size_t e = 0; e < ns; ++e) sum += (double)y[e];
    FILE* f = fopen(out_path, "wb");
    if (!f || fwrite(y, sizeof(float), ns, f) != ns) return 4;
    fclose(f);
    uint64_t nf = 0;
    CHECK_TIP(tip_forward_count(h, &nf));
    printf("c_host: B=%d T=%d y[0]=%.9g sum=%.9g forwards=%llu\n", B, T, (double)y[0], sum, (unsigned long long)nf);
    tip_destroy(h);
    return 0;
}
