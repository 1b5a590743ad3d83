// tip_abi.hip — the C-ABI of libtip_hip.so (include/tip_hip.h): handle, packed weight image, workspace
// carve-up and the forward orchestration.  No torch types, no exceptions across the boundary.
#include <math.h>
#include <string.h>

#include <new>

#include <mutex>
#include "tip_internal.h"
#include <atomic>
#include <chrono>

using namespace tip;

namespace {

const char* kStatusText[] = {
    "ok",
    "invalid argument",
    "unsupported configuration",
    "not ready: no packed weights attached",
    "workspace too small or misaligned",
    "HIP runtime error",
    "no HIP device",
    "allocation failure",
    "an inter-workgroup hand-off wait gave up in an earlier launch (outputs of that launch are NaN-poisoned); tip_check(h, 1) clears",
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// packed image builder: offsets are in floats, every section 64-float (256 B) aligned
struct Carver {
    size_t off = 0;
    size_t take(size_t n) {
        size_t o = off;
        off = align_up(off + n, 64);
        return o;
    }
};

PackedLinear carve_linear(Carver& c, int N, int K) {
    PackedLinear p;
    p.N = N;
    p.K = K;
    p.Npad = round_up(N, kGemmBN);
    p.Kpad = round_up(K, kGemmBK);
    p.w_off = c.take((size_t)p.Npad * p.Kpad);
    p.b_off = c.take((size_t)p.Npad);
    return p;
}

void build_layout(tip_handle* h) {
    const Dims& d = h->d;
    Carver c;
    PackedLayout& L = h->lay;
    L.in_lin = carve_linear(c, d.D, d.In);
    L.layers.assign(d.L, PackedLayer{});
    for (int l = 0; l < d.L; ++l) {
        PackedLayer& pl = L.layers[l];
        pl.qkv = carve_linear(c, 3 * d.D, d.D);
        pl.out = carve_linear(c, d.D, d.D);
        pl.ff1 = carve_linear(c, d.F, d.D);
        pl.ff2 = carve_linear(c, d.D, d.F);
        pl.g1_off = c.take(d.D);
        pl.be1_off = c.take(d.D);
        pl.g2_off = c.take(d.D);
        pl.be2_off = c.take(d.D);
        // big linears also in MFMA fragment order for the panel GEMM (tip_fused2.hip, launch_pgemm)
        for (PackedLinear* p : {&pl.qkv, &pl.out, &pl.ff1, &pl.ff2})
            if (pgemm_shape_ok(1 << 20, p->N, p->K)) p->f_off = c.take((size_t)p->N * p->K);
    }
    if (d.with_rnn) {
        L.rnn_ih = carve_linear(c, d.R, d.D);
        L.whh_frag_off = c.take((size_t)d.R * d.R);
        L.out_lin = carve_linear(c, d.S, d.R);
        L.out_frag_off = c.take((size_t)round_up(d.S, 16) * d.R);
    } else {
        L.rnn_ih = PackedLinear{};
        L.whh_frag_off = 0;
        L.out_lin = carve_linear(c, d.S, d.D);
        L.out_frag_off = c.take((size_t)round_up(d.S, 16) * d.D);
    }
    L.fused_floats = fused_packed_floats(d);
    L.fused_off = c.take(L.fused_floats);
    L.total_floats = c.off;
}

void build_tensor_table(tip_handle* h) {
    const Dims& d = h->d;
    auto add = [&](const std::string& n, int r, int cdim) {
        h->tensor_names.push_back(n);
        h->tensor_shapes.push_back({r, cdim});
    };
    add("in_linear.weight", d.D, d.In);
    add("in_linear.bias", d.D, 0);
    for (int l = 0; l < d.L; ++l) {
        const std::string p = "tf_encode.layers." + std::to_string(l) + ".";
        add(p + "self_attn.in_proj_weight", 3 * d.D, d.D);
        add(p + "self_attn.in_proj_bias", 3 * d.D, 0);
        add(p + "self_attn.out_proj.weight", d.D, d.D);
        add(p + "self_attn.out_proj.bias", d.D, 0);
        add(p + "linear1.weight", d.F, d.D);
        add(p + "linear1.bias", d.F, 0);
        add(p + "linear2.weight", d.D, d.F);
        add(p + "linear2.bias", d.D, 0);
        add(p + "norm1.weight", d.D, 0);
        add(p + "norm1.bias", d.D, 0);
        add(p + "norm2.weight", d.D, 0);
        add(p + "norm2.bias", d.D, 0);
    }
    if (d.with_rnn) {
        add("rnn.weight_ih_l0", d.R, d.D);
        add("rnn.weight_hh_l0", d.R, d.R);
        add("rnn.bias_ih_l0", d.R, 0);
        add("rnn.bias_hh_l0", d.R, 0);
        add("linear.weight", d.S, d.R);
    } else {
        add("linear.weight", d.S, d.D);
    }
    add("linear.bias", d.S, 0);
}

void pack_linear(float* img, const PackedLinear& p, const float* W, const float* b) {
    float* w = img + p.w_off;
    memset(w, 0, sizeof(float) * (size_t)p.Npad * p.Kpad);
    for (int n = 0; n < p.N; ++n) memcpy(w + (size_t)n * p.Kpad, W + (size_t)n * p.K, sizeof(float) * p.K);
    float* bb = img + p.b_off;
    memset(bb, 0, sizeof(float) * p.Npad);
    if (b) memcpy(bb, b, sizeof(float) * p.N);
}

Workspace carve_workspace(const Dims& d, int B, int T) {
    Workspace w;
    const size_t M = (size_t)B * T;
    const size_t Mp = align_up(M, 16);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off = align_up(off + n, 64); return o; };
    int big = 3 * d.D;
    if (d.F > big) big = d.F;
    if (d.R > big) big = d.R;
    if (d.InPad > big) big = d.InPad;
    w.flow = take(fused_supported(d, 40) && fused_has_rnn_ih(d) ? latency_flow_flag_floats() : 0);   // first: offset 0 whatever B and T are
    w.xa = take(Mp * d.D);
    w.xb = take(Mp * d.D);
    w.big = take(Mp * big);
    w.att = take(Mp * d.D);
    w.hall = take(Mp * (d.with_rnn ? d.R : 1));
    w.flags = take(rnn_flag_words(B, T) + 64);
    w.lat = take(latency_supported(d, B, T) ? latency_workspace_floats(B, T) : 0);   // (the persistent variant uses the same buffers)
    // window-split / pair-split plans: partial-sum images the partner workgroups exchange.  Only batches whose workgroups can all be
    // resident use them (2 x ceil(B / 2) <= #CUs <= 256), so the section stops growing at 256 windows — and does not shrink beyond:
    // AUTO runs a batch of whole rounds + a remainder as two forwards over the SAME workspace, and a part must never need more than
    // the whole (until round 4 the section was sized for B <= 1024, 168 MB at B = 1024, and absent above: a 1064-window batch
    // whose 1024-window part asked for more than the whole got TIP_ERR_WORKSPACE).
    w.xchg = take(fused2_supported(d, T) ? fused2s_xchg_floats(B < 256 ? B : 256) : 0);
    w.total_bytes = off * sizeof(float);
    return w;
}

int fail_hip(tip_handle* h, hipError_t e, const char* where) {
    h->last_hip_error = std::string(where) + ": " + hipGetErrorString(e);
    return TIP_ERR_HIP;
}

// profile == 1: every stage; profile == 2: only the dominant stage of the active plan (cheap enough to leave on
// inside a timed region).  One event pair per launch, accumulated until tip_set_option(TIP_OPT_PROFILE, ...).
bool stage_is_dominant(const char* name) {
    return !strcmp(name, "fused_encoder") || !strcmp(name, "ffn1_gemm");
}

struct StageScope {
    tip_handle* h;
    hipStream_t s;
    hipEvent_t stop = nullptr;
    StageScope(tip_handle* hh, hipStream_t ss, const char* name) : h(hh), s(ss) {
        if (!h->profile) return;
        if (h->profile >= 2 && !stage_is_dominant(name)) return;
        // 3: every FOURTH forward only — an event pair is two barrier packets in the queue, ~7 us per step at B = 256 (0.624 vs 0.631 ms,
        // measured), which a timed region should not carry on every launch
        if (h->profile == 3 && (h->forward_count & 3u) != 0) return;
        StageTimer* t = nullptr;
        for (auto& x : h->timers)
            if (x.name == name) { t = &x; break; }
        if (!t) {
            h->timers.emplace_back();
            t = &h->timers.back();
            t->name = name;
        }
        if (t->used == t->pairs.size()) {
            if (t->pairs.size() >= kMaxTimerPairs) return;
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            t->pairs.push_back({a, b});
        }
        auto& pr = t->pairs[t->used++];
        (void)hipEventRecord(pr.first, s);
        stop = pr.second;
    }
    ~StageScope() {
        if (stop) (void)hipEventRecord(stop, s);
    }
};

}  // namespace

// y = epi(x W^T + b (+ res)) for one packed linear of the general plan: the panel GEMM on the fragment copy when the layer
// has one and the batch is big enough, else the LDS-tiled GEMM on the row-major copy
static hipError_t linear_gemm(const float* P, const tip::PackedLinear& p, const float* A, int lda, const float* res, int ldres,
                              float* C, int ldc, int M, int flags, hipStream_t s) {
    static int use_pg = -1;   // TIP_GENERAL_PGEMM=0 keeps the LDS-tiled kernel (measurement)
    if (use_pg < 0) use_pg = (tip_env("TIP_GENERAL_PGEMM") && tip_env("TIP_GENERAL_PGEMM")[0] == '0') ? 0 : 1;
    // (the panel kernel's epilogue moves 16 bytes per lane: bias / residual / output rows must be 16-byte aligned — they are for
    //  every buffer the library carves out itself)
    auto al16 = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
    if (use_pg && p.f_off && tip::pgemm_shape_ok(M, p.N, p.K) && al16(P + p.b_off, 0) && al16(C, ldc) && (!(flags & 2) || al16(res, ldres)))
        return tip::launch_pgemm(A, lda, P + p.f_off, (size_t)p.N * p.K, P + p.b_off, res, ldres, C, ldc, M, p.N, p.K, flags, s);
    return tip::launch_gemm(A, lda, P + p.w_off, p.Kpad, P + p.b_off, res, ldres, C, ldc, M, p.N, p.Npad, flags, s);
}

namespace tip {
namespace {
struct DevSerialState {
    std::mutex mu;
    hipStream_t last = nullptr;
    bool have = false;
    hipEvent_t ev = nullptr;
};
DevSerialState g_serial[kMaxDevices];
}  // namespace

CoopSerial::CoopSerial(int device, hipStream_t s)
    : dev(device >= 0 && device < kMaxDevices ? device : tip_cur_device()), stream(s), capturing(stream_is_capturing(s)), status(hipSuccess) {
    DevSerialState& st = g_serial[dev];
    st.mu.lock();
    if (capturing) return;   // see stream_is_capturing (tip_internal.h)
    // The event is recorded at the END of every forward (destructor), on the stream that forward ran on — known to be alive then.  A
    // stream switch therefore only waits on the event: it never touches the previous stream's handle, which may have been destroyed
    // (or recycled) by now (hipEventRecord on a dangling hipStream_t is undefined behaviour), and it cannot pull this
    // stream into another stream's capture.
    if (st.have && st.last != s && st.ev) status = hipStreamWaitEvent(s, st.ev, 0);
}

CoopSerial::~CoopSerial() {
    DevSerialState& st = g_serial[dev];
    if (!capturing) {
        if (!st.ev && hipEventCreateWithFlags(&st.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            st.ev = nullptr;
        }
        if (st.ev) (void)hipEventRecord(st.ev, stream);
        st.last = stream;
        st.have = true;
    }
    st.mu.unlock();
}
}  // namespace tip

extern "C" {

int tip_abi_version(void) { return TIP_ABI_VERSION; }

const char* tip_strerror(int status) {
    const int i = -status;
    if (i < 0 || i >= (int)(sizeof(kStatusText) / sizeof(kStatusText[0]))) return "unknown status";
    return kStatusText[i];
}

int tip_create(const tip_config* cfg, tip_handle** out) {
    if (!cfg || !out) return TIP_ERR_INVALID_ARG;
    *out = nullptr;
    Dims d{};
    d.n_imu_total = cfg->input_size_imu + (cfg->with_acc_sum ? 18 : 0);  // simple_transformer_with_state.py:20-24
    d.S = cfg->size_s;
    d.In = d.n_imu_total + d.S;
    d.InPad = round_up(d.In, kGemmBK);
    d.D = cfg->tf_in_dim;
    d.H = cfg->n_heads;
    d.F = cfg->tf_hid_size;
    d.L = cfg->tf_layers;
    d.R = cfg->with_rnn ? cfg->rnn_hid_size : 0;
    d.with_rnn = cfg->with_rnn ? 1 : 0;
    d.t_max = cfg->t_max > 0 ? cfg->t_max : 40;
    d.rootv0 = 18 * 6;       // :75
    d.rootv1 = 18 * 6 + 3;
    if (cfg->input_size_imu <= 0 || d.S <= 0 || d.D <= 0 || d.H <= 0 || d.F <= 0 || d.L < 0) return TIP_ERR_INVALID_ARG;
    if (d.D % d.H) return TIP_ERR_INVALID_ARG;  // nn.MultiheadAttention asserts the same
    d.dh = d.D / d.H;
    if (d.D % 16 || d.F % 16 || (d.with_rnn && (d.R % 64)) ||
        !(d.dh == 8 || d.dh == 16 || d.dh == 32 || d.dh == 64) || d.D > 2048 || d.S < d.rootv1)
        return TIP_ERR_UNSUPPORTED_CONFIG;
    if (d.with_rnn && ((d.R / 16) + 3) / 4 > 8) return TIP_ERR_UNSUPPORTED_CONFIG;
    // 1/sqrt(dh) is a power of two for dh in {16, 64}: fold it into W_q / b_q exactly
    const float sc = 1.0f / sqrtf((float)d.dh);
    d.fold_q_scale = (d.dh == 16 || d.dh == 64) ? 1 : 0;
    d.q_scale = d.fold_q_scale ? 1.0f : sc;

    tip_handle* h = new (std::nothrow) tip_handle();
    if (!h) return TIP_ERR_ALLOC;
    h->cfg = *cfg;
    h->d = d;
    build_tensor_table(h);
    build_layout(h);
    h->fuse_head = 0;   // TIP_OPT_FUSE_HEAD: reserved (accepted, no effect)
    {
        // base of the dataflow launches' epoch stamps: different per handle and per process start, so that flag words another handle
        // (or an earlier life of the same address) left in a caller's workspace never read as this handle's (48 bits of splitmix64)
        static std::atomic<unsigned long long> serial{0};
        unsigned long long z = (unsigned long long)reinterpret_cast<uintptr_t>(h) ^ (serial.fetch_add(1) << 40) ^
                               (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
        z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        h->flow_epoch = z & 0x0000ffffffffffffull;
    }
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0) {
        h->device = dev;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            h->num_cus = prop.multiProcessorCount;
        // the hand-off error word: pinned host memory the kernels can store to and the host can read without a sync
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess && hp) {
            memset(hp, 0, 64);
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess && dp) {
                h->err_host = static_cast<unsigned*>(hp);
                h->err_dev = static_cast<unsigned*>(dp);
            } else {
                (void)hipHostFree(hp);
            }
        }
        (void)hipGetLastError();
    } else {
        (void)hipGetLastError();
    }
    *out = h;
    return TIP_OK;
}

void tip_destroy(tip_handle* h) {
    if (!h) return;
    for (auto& t : h->timers)
        for (auto& pr : t.pairs) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    if (h->err_host) (void)hipHostFree(h->err_host);
    delete h;
}

const char* tip_last_hip_error(const tip_handle* h) { return h ? h->last_hip_error.c_str() : ""; }

int tip_set_option(tip_handle* h, int option, int value) {
    if (!h) return TIP_ERR_INVALID_ARG;
    switch (option) {
        case TIP_OPT_PLAN:
            if (value < TIP_PLAN_AUTO || value > TIP_PLAN_FUSED1S || value == 9 /* reserved */) return TIP_ERR_INVALID_ARG;
            if (value == 5 || value == 7 || value == 8) return TIP_ERR_UNSUPPORTED_CONFIG;   // retired plans (ABI 4): pair-split, split-fp16
            h->plan = value;
            return TIP_OK;
        case TIP_OPT_PROFILE:
            if (value < 0 || value > 3) return TIP_ERR_INVALID_ARG;
            h->profile = value;
            for (auto& t : h->timers) t.used = 0;  // reset the accumulators
            return TIP_OK;
        case TIP_OPT_RNN_CLUSTER:
            if (!(value == 0 || value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32 || value == TIP_RNN_CLUSTER_ROWS4))
                return TIP_ERR_INVALID_ARG;
            h->rnn_cluster = value;
            return TIP_OK;
        case TIP_OPT_FAULT_INJECT:
            if (value < 0 || value > 31) return TIP_ERR_INVALID_ARG;
            h->fault_inject = value;
            return TIP_OK;
        case TIP_OPT_FUSE_HEAD:
            if (value < 0 || value > 1) return TIP_ERR_INVALID_ARG;
            h->fuse_head = value;
            return TIP_OK;
        case TIP_OPT_AUTO_DEMOTE:
            if (value < 0 || value > 1) return TIP_ERR_INVALID_ARG;
            h->auto_demote = value;
            return TIP_OK;
        case TIP_OPT_DEMOTED:
            if (value < 0 || value > 1) return TIP_ERR_INVALID_ARG;
            h->demoted = value;
            return TIP_OK;
        case TIP_OPT_F1S_PARTS:
            if (value != 0 && value != 2 && value != 4) return TIP_ERR_INVALID_ARG;
            h->f1s_parts = value;
            return TIP_OK;
        case TIP_OPT_NO_FLOW:
            if (value < 0 || value > 1) return TIP_ERR_INVALID_ARG;
            h->no_flow = value;
            return TIP_OK;
        default: return TIP_ERR_INVALID_ARG;
    }
}

int tip_check(tip_handle* h, int clear) {
    if (!h) return TIP_ERR_INVALID_ARG;
    if (!h->err_host) return TIP_OK;
    volatile unsigned* w = const_cast<volatile unsigned*>(h->err_host);
    const unsigned v = w[0] | w[1];
    if (clear) w[0] = w[1] = 0u;
    return v ? TIP_ERR_HANDOFF : TIP_OK;
}

int tip_get_option(const tip_handle* h, int option, int* value) {
    if (!h || !value) return TIP_ERR_INVALID_ARG;
    switch (option) {
        case TIP_OPT_PLAN: *value = h->plan; return TIP_OK;
        case TIP_OPT_PROFILE: *value = h->profile; return TIP_OK;
        case TIP_OPT_RNN_CLUSTER: *value = h->rnn_cluster; return TIP_OK;
        case TIP_OPT_FAULT_INJECT: *value = h->fault_inject; return TIP_OK;
        case TIP_OPT_FUSE_HEAD: *value = h->fuse_head; return TIP_OK;
        case TIP_OPT_AUTO_DEMOTE: *value = h->auto_demote; return TIP_OK;
        case TIP_OPT_DEMOTED: *value = h->demoted; return TIP_OK;
        case TIP_OPT_F1S_PARTS: *value = h->f1s_parts; return TIP_OK;
        case TIP_OPT_NO_FLOW: *value = h->no_flow; return TIP_OK;
        case TIP_OPT_HANDOFF_KIND: {
            const volatile unsigned* w = const_cast<const volatile unsigned*>(h->err_host);
            *value = !w ? 0 : w[0] ? 1 : w[1] ? 2 : 0;
            return TIP_OK;
        }
        default: return TIP_ERR_INVALID_ARG;
    }
}

int tip_num_tensors(const tip_handle* h) { return h ? (int)h->tensor_names.size() : TIP_ERR_INVALID_ARG; }

int tip_tensor_info(const tip_handle* h, int i, const char** name, int* rows, int* cols) {
    if (!h || i < 0 || i >= (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (name) *name = h->tensor_names[i].c_str();
    if (rows) *rows = h->tensor_shapes[i].first;
    if (cols) *cols = h->tensor_shapes[i].second;
    return TIP_OK;
}

int tip_packed_bytes(const tip_handle* h, size_t* bytes) {
    if (!h || !bytes) return TIP_ERR_INVALID_ARG;
    *bytes = h->lay.total_floats * sizeof(float);
    return TIP_OK;
}

int tip_pack_weights(const tip_handle* h, const float* const* t, int n, void* packed_host_out, size_t bytes) {
    if (!h || !t || !packed_host_out) return TIP_ERR_INVALID_ARG;
    if (n != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (bytes < h->lay.total_floats * sizeof(float)) return TIP_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!t[i]) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    const PackedLayout& L = h->lay;
    float* img = static_cast<float*>(packed_host_out);
    memset(img, 0, L.total_floats * sizeof(float));

    // in_linear with the channel shuffle (:88-89) folded into its rows and the root-velocity zeroing (:75)
    // folded into its columns: packed row a*H + b = reference row b*dh + a.
    {
        const PackedLinear& p = L.in_lin;
        float* w = img + p.w_off;
        float* b = img + p.b_off;
        for (int a = 0; a < d.dh; ++a)
            for (int hb = 0; hb < d.H; ++hb) {
                const int nn = a * d.H + hb, old = hb * d.dh + a;
                memcpy(w + (size_t)nn * p.Kpad, t[0] + (size_t)old * d.In, sizeof(float) * d.In);
                for (int c = d.rootv0; c < d.rootv1; ++c) w[(size_t)nn * p.Kpad + d.n_imu_total + c] = 0.f;
                b[nn] = t[1][old];
            }
    }
    for (int l = 0; l < d.L; ++l) {
        const float* const* lw = t + 2 + 12 * l;
        const PackedLayer& pl = L.layers[l];
        pack_linear(img, pl.qkv, lw[0], lw[1]);
        if (d.fold_q_scale) {
            const float sc = 1.0f / sqrtf((float)d.dh);  // power of two: exact
            float* w = img + pl.qkv.w_off;
            float* b = img + pl.qkv.b_off;
            for (size_t i = 0; i < (size_t)d.D * pl.qkv.Kpad; ++i) w[i] *= sc;
            for (int i = 0; i < d.D; ++i) b[i] *= sc;
        }
        pack_linear(img, pl.out, lw[2], lw[3]);
        pack_linear(img, pl.ff1, lw[4], lw[5]);
        pack_linear(img, pl.ff2, lw[6], lw[7]);
        for (const PackedLinear* p : {&pl.qkv, &pl.out, &pl.ff1, &pl.ff2}) {
            if (!p->f_off) continue;
            // from the packed row-major copy (which already carries the folds) to [N/16][K/16][64 lanes][4]
            const float* w = img + p->w_off;
            float* f = img + p->f_off;
            const int KBn = p->K / 16;
            for (int nb = 0; nb < p->N / 16; ++nb)
                for (int kb = 0; kb < KBn; ++kb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int s4 = 0; s4 < 4; ++s4)
                            f[((size_t)(nb * KBn + kb) * 64 + lane) * 4 + s4] =
                                w[(size_t)(nb * 16 + (lane & 15)) * p->Kpad + kb * 16 + 4 * (lane >> 4) + s4];
        }
        memcpy(img + pl.g1_off, lw[8], sizeof(float) * d.D);
        memcpy(img + pl.be1_off, lw[9], sizeof(float) * d.D);
        memcpy(img + pl.g2_off, lw[10], sizeof(float) * d.D);
        memcpy(img + pl.be2_off, lw[11], sizeof(float) * d.D);
    }
    const float* const* tw = t + 2 + 12 * d.L;
    if (d.with_rnn) {
        pack_linear(img, L.rnn_ih, tw[0], tw[2]);
        float* b = img + L.rnn_ih.b_off;
        for (int i = 0; i < d.R; ++i) b[i] = tw[2][i] + tw[3][i];  // b_ih + b_hh
        const int KB = d.R / 16;
        float* f = img + L.whh_frag_off;
        for (int nb = 0; nb < KB; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s)
                        f[((size_t)(nb * KB + kb) * 64 + lane) * 4 + s] =
                            tw[1][(size_t)(nb * 16 + (lane & 15)) * d.R + kb * 16 + 4 * (lane >> 4) + s];
        pack_linear(img, L.out_lin, tw[4], tw[5]);
    } else {
        pack_linear(img, L.out_lin, tw[0], tw[1]);
    }
    {
        // out-linear in fragment order for the skinny head GEMM
        const int K = d.with_rnn ? d.R : d.D, KB = K / 16, NBo = round_up(d.S, 16) / 16;
        const float* Wo = d.with_rnn ? tw[4] : tw[0];
        float* f = img + L.out_frag_off;
        for (int nb = 0; nb < NBo; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int n = nb * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + s4;
                        f[((size_t)(nb * KB + kb) * 64 + lane) * 4 + s4] = n < d.S ? Wo[(size_t)n * K + k] : 0.f;
                    }
    }
    if (L.fused_floats) fused_pack(d, t, img + L.fused_off);
    return TIP_OK;
}

// The same image built on the GPU from device tensors (tip_pack.hip); `packed_dev` is then ready for tip_attach_packed.
int tip_pack_weights_device(const tip_handle* h, const float* const* t, int n, void* packed_dev, size_t bytes, void* stream) {
    if (!h || !t || !packed_dev) return TIP_ERR_INVALID_ARG;
    if (n != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (bytes < h->lay.total_floats * sizeof(float)) return TIP_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!t[i]) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    const PackedLayout& L = h->lay;
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<PackOp> ops;
    auto op = [](const float* src, size_t dst, int N, int K, int src_rows, int src_cols, int frag) {
        PackOp o;
        o.src = src; o.src2 = nullptr; o.dst_off = dst; o.N = N; o.K = K; o.src_rows = src_rows; o.src_cols = src_cols; o.frag = frag;
        o.shuffle_h = 0; o.shuffle_dh = 0; o.z0 = 0; o.z1 = 0; o.scale_rows = 0; o.scale = 1.f; o.transpose = 0;
        return o;
    };
    auto linear = [&](const PackedLinear& p, const float* W, const float* b) {
        ops.push_back(op(W, p.w_off, p.Npad, p.Kpad, p.N, p.K, 0));
        ops.push_back(op(b, p.b_off, p.Npad, 1, p.N, 1, 0));
    };
    {
        PackOp w = op(t[0], L.in_lin.w_off, L.in_lin.Npad, L.in_lin.Kpad, d.D, d.In, 0);
        w.shuffle_h = d.H; w.shuffle_dh = d.dh;
        w.z0 = d.n_imu_total + d.rootv0; w.z1 = d.n_imu_total + d.rootv1;
        ops.push_back(w);
        PackOp b = op(t[1], L.in_lin.b_off, L.in_lin.Npad, 1, d.D, 1, 0);
        b.shuffle_h = d.H; b.shuffle_dh = d.dh;
        ops.push_back(b);
    }
    for (int l = 0; l < d.L; ++l) {
        const float* const* lw = t + 2 + 12 * l;
        const PackedLayer& pl = L.layers[l];
        linear(pl.qkv, lw[0], lw[1]);
        if (d.fold_q_scale) {
            const float sc = 1.0f / sqrtf((float)d.dh);
            ops[ops.size() - 2].scale = sc; ops[ops.size() - 2].scale_rows = d.D;
            ops[ops.size() - 1].scale = sc; ops[ops.size() - 1].scale_rows = d.D;
        }
        linear(pl.out, lw[2], lw[3]);
        linear(pl.ff1, lw[4], lw[5]);
        linear(pl.ff2, lw[6], lw[7]);
        {
            const PackedLinear* pls[4] = {&pl.qkv, &pl.out, &pl.ff1, &pl.ff2};
            const float* src[4] = {lw[0], lw[2], lw[4], lw[6]};
            for (int i = 0; i < 4; ++i) {
                if (!pls[i]->f_off) continue;
                PackOp f = op(src[i], pls[i]->f_off, pls[i]->N, pls[i]->K, pls[i]->N, pls[i]->K, 1);
                if (i == 0 && d.fold_q_scale) {
                    f.scale = 1.0f / sqrtf((float)d.dh);
                    f.scale_rows = d.D;
                }
                ops.push_back(f);
            }
        }
        ops.push_back(op(lw[8], pl.g1_off, d.D, 1, d.D, 1, 0));
        ops.push_back(op(lw[9], pl.be1_off, d.D, 1, d.D, 1, 0));
        ops.push_back(op(lw[10], pl.g2_off, d.D, 1, d.D, 1, 0));
        ops.push_back(op(lw[11], pl.be2_off, d.D, 1, d.D, 1, 0));
    }
    const float* const* tw = t + 2 + 12 * d.L;
    const float* Wo;
    int Ko;
    if (d.with_rnn) {
        linear(L.rnn_ih, tw[0], tw[2]);
        ops.back().src2 = tw[3];                                           // b_ih + b_hh
        ops.push_back(op(tw[1], L.whh_frag_off, d.R, d.R, d.R, d.R, 1));
        linear(L.out_lin, tw[4], tw[5]);
        Wo = tw[4]; Ko = d.R;
    } else {
        linear(L.out_lin, tw[0], tw[1]);
        Wo = tw[0]; Ko = d.D;
    }
    ops.push_back(op(Wo, L.out_frag_off, round_up(d.S, 16), Ko, d.S, Ko, 1));
    if (L.fused_floats) fused_pack_ops(d, t, L.fused_off, ops);
    if (hipMemsetAsync(packed_dev, 0, L.total_floats * sizeof(float), s) != hipSuccess) return TIP_ERR_HIP;
    if (run_pack_ops(ops, static_cast<float*>(packed_dev), s) != hipSuccess) return TIP_ERR_HIP;
    return TIP_OK;
}

int tip_attach_packed(tip_handle* h, const void* packed_device, size_t bytes) {
    if (!h || !packed_device) return TIP_ERR_INVALID_ARG;
    if (bytes < h->lay.total_floats * sizeof(float)) return TIP_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(packed_device) % 256) return TIP_ERR_INVALID_ARG;
    h->packed_dev = static_cast<const float*>(packed_device);
    return TIP_OK;
}

int tip_workspace_bytes(const tip_handle* h, int B, int T, size_t* bytes) {
    if (!h || !bytes || B < 0 || T < 1) return TIP_ERR_INVALID_ARG;
    *bytes = carve_workspace(h->d, B, T).total_bytes + 256;
    return TIP_OK;
}

int tip_max_batch(const tip_handle* h, int T, int fp64, int* max_batch) {
    if (!h || !max_batch || T < 1) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    long long m;
    if (fp64) {
        // tip_forward_f64: element offsets of the widest activation and of the attention grid in 32 bits; 64-row GEMM tiles on grid.y
        const long long widest = 3 * d.D > d.F ? 3 * d.D : d.F;
        m = 0x7fffffffLL / (widest * T);
        m = std::min(m, 0x7fffffffLL / ((long long)T * d.H));
        m = std::min(m, 65535LL * 64 / T);
    } else {
        // tip_forward: byte offsets of the widest activation row block through 32-bit buffer descriptors
        const long long widest = std::max(std::max(3 * d.D, d.F), std::max(d.R, d.InPad));
        m = 0x7fffffffLL / (4 * widest * T);
    }
    *max_batch = (int)std::min<long long>(m, 0x7fffffff);
    return TIP_OK;
}

int tip_forward_count(const tip_handle* h, uint64_t* n) {
    if (!h || !n) return TIP_ERR_INVALID_ARG;
    *n = h->forward_count;
    return TIP_OK;
}

int tip_profile_read(tip_handle* h, const char** names, float* ms, int* launches, int cap) {
    if (!h || cap < 0) return TIP_ERR_INVALID_ARG;
    int n = 0;
    for (auto& t : h->timers) {
        if (!t.used) continue;
        if (n < cap) {
            double total = 0.0;
            for (size_t i = 0; i < t.used; ++i) {
                float v = 0.f;
                hipError_t e = hipEventElapsedTime(&v, t.pairs[i].first, t.pairs[i].second);
                if (e != hipSuccess) return fail_hip(h, e, "hipEventElapsedTime");
                total += v;
            }
            if (names) names[n] = t.name.c_str();
            if (ms) ms[n] = (float)total;
            if (launches) launches[n] = (int)t.used;
            ++n;
        }
    }
    return n;
}

int tip_spin_timeouts(unsigned* count) {
    if (!count) return TIP_ERR_INVALID_ARG;
    unsigned a = 0, b = 0, c = 0;
    if (read_spin_timeouts_general(&a) != hipSuccess || read_spin_timeouts_latency(&b) != hipSuccess ||
        read_spin_timeouts_fused2(&c) != hipSuccess)
        return TIP_ERR_HIP;
    *count = a + b + c;
    return TIP_OK;
}

// The ring of tip_forward_reuse, when the forward below is that entry point's (null: tip_forward)
struct ReuseCtx {
    float* cache;
    const int* frame_ctr;
    int frame_idx;
};
static int forward_impl(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                        const float* keep_mask, float keep_scale, void* workspace, size_t workspace_bytes,
                        tip_stream_t stream, const ReuseCtx* reuse);

int tip_forward(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                const float* keep_mask, float keep_scale, void* workspace, size_t workspace_bytes,
                tip_stream_t stream) {
    return forward_impl(h, x_imu, x_s, y, B, T, flags, keep_mask, keep_scale, workspace, workspace_bytes, stream, nullptr);
}

int tip_reuse_cache_bytes(const tip_handle* h, int n_streams, size_t* bytes) {
    if (!h || !bytes || n_streams < 0) return TIP_ERR_INVALID_ARG;
    if (!fused2_supported(h->d, 40)) return TIP_ERR_UNSUPPORTED_CONFIG;
    *bytes = reuse_cache_floats(n_streams) * sizeof(float);
    return TIP_OK;
}

int tip_reuse_reset(void* cache, size_t cache_bytes, tip_stream_t stream) {
    if (!cache || cache_bytes < reuse_cache_floats(0) * sizeof(float)) return TIP_ERR_INVALID_ARG;
    // every slot's tag <- INT_MIN: no frame index a caller may pass (>= 0) makes a window's 40 tags match
    return hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cache), (int)0x80000000, reuse_cache_floats(0), static_cast<hipStream_t>(stream)) == hipSuccess
               ? TIP_OK
               : TIP_ERR_HIP;
}

int tip_forward_reuse(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags, void* cache,
                      size_t cache_bytes, int frame_idx, const int* frame_ctr, void* workspace, size_t workspace_bytes,
                      tip_stream_t stream) {
    if (!h || !cache || B < 0 || T < 1 || (flags & TIP_FWD_KEEP_MASK) || (!frame_ctr && frame_idx < 0)) return TIP_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(cache) % 256 || cache_bytes < reuse_cache_floats(B) * sizeof(float)) return TIP_ERR_WORKSPACE;
    if (!fused2_supported(h->d, 40) || T > 40) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (T == 40 && !frame_ctr && frame_idx < 39) return TIP_ERR_INVALID_ARG;   // a full window has 39 earlier frames
    const ReuseCtx rc{static_cast<float*>(cache), frame_ctr, frame_idx};
    return forward_impl(h, x_imu, x_s, y, B, T, flags, nullptr, 1.f, workspace, workspace_bytes, stream, &rc);
}

static int forward_impl(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                        const float* keep_mask, float keep_scale, void* workspace, size_t workspace_bytes,
                        tip_stream_t stream, const ReuseCtx* reuse) {
    if (!h || !x_imu || !x_s || !y || B < 0 || T < 1) return TIP_ERR_INVALID_ARG;
    // (cfg.t_max is a sizing hint, not a limit: the reference builds its causal mask for any window length, :56-58,85; what
    // bounds B * T here are the 32-bit byte offsets of the buffer descriptors)
    if ((long long)B * T * (long long)std::max(std::max(3 * h->d.D, h->d.F), std::max(h->d.R, h->d.InPad)) * 4 > 0x7fffffffLL)
        return TIP_ERR_UNSUPPORTED_CONFIG;
    if ((flags & TIP_FWD_KEEP_MASK) && !keep_mask) return TIP_ERR_INVALID_ARG;
    if (!h->packed_dev) return TIP_ERR_NOT_READY;
    if (tip_check(h, 0) != TIP_OK) return TIP_ERR_HANDOFF;   // sticky: an earlier launch lost a hand-off (tip_check(h, 1) clears)
    if (B == 0) return TIP_OK;
    const Dims& d = h->d;
    const Workspace ws = carve_workspace(d, B, T);
    const Guard gd = h->guard();
    if (!workspace || reinterpret_cast<uintptr_t>(workspace) % 256 || workspace_bytes < ws.total_bytes)
        return TIP_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cus = effective_cus(h->num_cus, s);   // the stream's CU mask counts, not the device's CU total
    // AUTO, a batch that is whole rounds of #CUs windows plus a SMALL remainder: the one-window kernel takes a full round (0.53 ms +
    // the tail) for the remainder alone; the few-stream latency plan takes 0.16-0.28 ms for up to 32 windows and the window-split
    // encoder 0.30 ms for up to #CUs / 4 (one window on four CUs), 0.45 ms for up to #CUs / 2.  Run the whole rounds
    // and the remainder as two launch sequences when the model below says so (stream-ordered: they share the workspace); every
    // window's result is bit-identical to what its part's plan gives on its own (tests/test_benchmarked_shapes_gpu.py).
    // Costs in us from profiles/r04/plan_bench_split.txt (B = 256 step 0.625 ms; the remainder's latency-plan forward measured
    // 163 / 177 / 201 / 282 / 372 us for 1 / 8 / 16 / 32 / 44 windows behind it); TIP_AUTO_SPLIT=0 disables (measurement).
    // (the cost model below is calibrated at T = 40 — the only window length the window-split plans serve — and its constants scale
    // with the CU count only through `cus`, which is the device's: other window lengths take whole rounds)
    // (tip_forward_reuse, full windows: ONE launch sequence on the two-window encoder's reuse form, whatever the batch)
    const bool reuse_full = reuse && T == 40;
    if (!reuse_full && h->plan == TIP_PLAN_AUTO && !h->demoted && cus == h->num_cus && B > cus && T == 40 && fused_supported(d, T) && fused_has_rnn_ih(d)) {
        static const bool split_on = !(tip_env("TIP_AUTO_SPLIT") && tip_env("TIP_AUTO_SPLIT")[0] == '0');
        const int r = B % cus, bm = B - r;
        // what the remainder costs on its own (us; AUTO's choice for that many windows, below): the latency plan up to 32 windows
        // (measured 163 / 177 / 201 / 282 / 372 us for 1 / 8 / 16 / 32 / 44), the window-split encoder up to #CUs / 4 (0.305 ms) and
        // #CUs / 2 (0.45 ms)
        long long rem = -1;
        const bool quad = fused2_supported(d, T) && h->f1s_parts != 2 && fused1s_quad_fits(r, cus);
        // (round 6, the few-stream plan as one launch: 147 / 149 / 181 / 241 us for <= 1 / 8 / 16 / 24 windows — a step per window that shares
        //  an XCD — and the launch chain's 286 at 32, 372 at 44; tools/auto_calibrate.py --stages re-measures every constant here)
        auto lat_us = [](int n) { return n <= 8 ? 147LL + n / 4 : n <= 16 ? 181LL : n <= 24 ? 241LL : n <= 32 ? 200LL + (long long)(3.6 * (n - 8)) : 286LL + 10LL * (n - 32); };
        if (r >= 1 && r <= (quad ? 32 : 48) && latency_supported(d, r, T)) rem = lat_us(r);
        else if (r >= 1 && fused2_supported(d, T) && fused1s_fits(r, cus)) rem = quad ? 305 : 452;
        else if (r >= 1 && latency_supported(d, r, T)) rem = lat_us(r);
        if (split_on && rem >= 0) {
            auto single = [&](long long b) {   // encoder rounds of the cheaper of the two fused kernels + recurrence / projection rounds
                const long long rh = (b + cus - 1) / cus, r2 = ((b + 1) / 2 + cus - 1) / cus;
                const long long enc = (fused2_supported(d, T) && r2 * 1049 < rh * 527) ? r2 * 1049 : rh * 527;
                return enc + 96 * rh;
            };
            // (a part never needs more workspace than the whole — carve_workspace is monotone, tests/test_host_cpu.py — but a caller's
            // buffer sized by an older library must fall through to the single launch sequence, not fail)
            // Round 5: a remainder that takes the window-split encoder shares ONE recurrence and ONE output projection with the whole
            // rounds — the two encoders write their windows' input terms (and arm their HALL rows) side by side in the whole batch's
            // workspace — instead of bringing a 62-us recurrence + a projection launch of its own: the recurrence serves the extra
            // tiles next to the ones it has (their hops overlap), ~35 us for up to 128 more windows.  rnn_hidden 512 only (the
            // four-window recurrence, whose per-window results do not depend on the tiling); bit-identical to the two sequences.
            const bool rem_f1s = r >= 1 && !(r <= (quad ? 32 : 48) && latency_supported(d, r, T)) && fused2_supported(d, T) && fused1s_fits(r, cus);
            static const bool merge_on = !(tip_env("TIP_AUTO_MERGE") && tip_env("TIP_AUTO_MERGE")[0] == '0');   // measurement
            // (the recurrence advances 1, 2 or 4 tiles per cluster together: a third tile costs a fourth's time, and a fifth a second
            // pass — B = 556 measured 1 533 us merged against 1 509: merge only where the remainder does not push the whole rounds'
            // tile count per cluster across such a step, or the rounds have a single tile)
            const int tpg_b = ((B + 3) / 4 + 63) / 64, tpg_m = ((bm + 3) / 4 + 63) / 64;
            const bool tiles_ok = cus == 256 && (tpg_b <= 2 || (tpg_b <= 4 && tpg_m >= 3));
            if (merge_on && rem_f1s && tiles_ok && d.with_rnn && d.R == 512 && h->rnn_cluster == 0 && single(bm) + (quad ? 232 : 375) + 35 < single(B)) {
                CoopSerial serial(h->device, s);
                if (serial.status != hipSuccess) return fail_hip(h, serial.status, "stream serialisation");
                const float* P = h->packed_dev;
                const PackedLayout& L = h->lay;
                float* W0 = static_cast<float*>(workspace);
                float* big = W0 + ws.big;
                float* hall = W0 + ws.hall;
                const float* mask = (flags & TIP_FWD_KEEP_MASK) ? keep_mask : nullptr;
                const float ks = mask ? keep_scale : 1.f;
                const size_t row_i = (size_t)T * d.n_imu_total, row_s = (size_t)T * d.S, row_r = (size_t)T * d.R;
                const bool armed = rnn_uses_sentinel(d, B, T, kRnnRows4);
                hipError_t e;
                {
                    StageScope sc(h, s, "fused_encoder");
                    const long long cusl = cus, rounds_h = (bm + cusl - 1) / cusl, rounds_2 = ((bm + 1) / 2 + cusl - 1) / cusl;
                    if (fused2_supported(d, T) && rounds_2 * 1049 < rounds_h * 527)
                        e = launch_fused_encoder2(d, P + L.fused_off, x_imu, x_s, mask, ks, big, armed ? hall : nullptr, bm, cus, s);
                    else
                        e = launch_fused_encoder_h(d, P + L.fused_off, x_imu, x_s, mask, ks, nullptr, big, armed ? hall : nullptr, bm, T, cus, s);
                    if (e != hipSuccess) return fail_hip(h, e, "fused_encoder");
                }
                {
                    StageScope sc(h, s, "fused_encoder");   // (the remainder's encoder: a stage of its own in the profile, as in the two-sequence form)
                    e = launch_fused_encoder1s(d, P + L.fused_off, x_imu + bm * row_i, x_s + bm * row_s, mask ? mask + bm * row_s : nullptr, ks,
                                               big + bm * row_r, armed ? hall + bm * row_r : nullptr, W0 + ws.xchg, r, cus, h->f1s_parts, gd, s);
                    if (e != hipSuccess) return fail_hip(h, e, "fused_encoder1s");
                }
                {
                    StageScope sc(h, s, "rnn_recurrence");
                    e = launch_rnn(d, big, P + L.whh_frag_off, hall, reinterpret_cast<unsigned*>(W0 + ws.flags), B, T, kRnnRows4, cus, armed, gd, s);
                    if (e != hipSuccess) return fail_hip(h, e, "rnn_recurrence");
                }
                {
                    StageScope sc(h, s, "out_linear");
                    const bool last_only = (flags & TIP_FWD_LAST_ROW_ONLY) != 0;
                    const float* hA = last_only ? hall + (size_t)(T - 1) * d.R : hall;
                    const long long hlda = last_only ? (long long)T * d.R : d.R;
                    const int hM = last_only ? B : B * T;
                    static const bool ksplit = !(tip_env("TIP_HEAD") && tip_env("TIP_HEAD")[0] == 'o');   // TIP_HEAD=old: measurement
                    e = hipErrorInvalidValue;
                    if (ksplit && T % 40 == 0)
                        e = launch_head_ksplit(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, hM, d.S, d.R, last_only, cus, s);
                    if (e == hipErrorInvalidValue)
                        e = launch_head_gemm(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, hM, d.S, d.R, s);
                    if (e != hipSuccess) return fail_hip(h, e, "out_linear");
                }
                h->forward_count++;
                return TIP_OK;
            }
            if (single(bm) + rem < single(B) && carve_workspace(d, bm, T).total_bytes <= workspace_bytes &&
                carve_workspace(d, r, T).total_bytes <= workspace_bytes) {
                const size_t row_i = (size_t)T * d.n_imu_total, row_s = (size_t)T * d.S;
                const size_t row_y = (flags & TIP_FWD_LAST_ROW_ONLY) ? (size_t)d.S : row_s;
                const uint64_t count0 = h->forward_count;
                int st = tip_forward(h, x_imu, x_s, y, bm, T, flags, keep_mask, keep_scale, workspace, workspace_bytes, stream);
                if (st != TIP_OK) return st;
                st = tip_forward(h, x_imu + bm * row_i, x_s + bm * row_s, y + bm * row_y, r, T, flags, keep_mask ? keep_mask + bm * row_s : nullptr,
                                 keep_scale, workspace, workspace_bytes, stream);
                if (st != TIP_OK) return st;
                h->forward_count = count0 + 1;   // one forward, two launch sequences
                return TIP_OK;
            }
        }
    }
    CoopSerial serial(h->device, s);   // forwards of different streams do not overlap on the device (cooperating kernels)
    if (serial.status != hipSuccess) return fail_hip(h, serial.status, "stream serialisation");
    const float* P = h->packed_dev;
    const PackedLayout& L = h->lay;
    float* W0 = static_cast<float*>(workspace);
    float* xa = W0 + ws.xa;
    float* xb = W0 + ws.xb;
    float* big = W0 + ws.big;
    float* att = W0 + ws.att;
    float* hall = W0 + ws.hall;
    unsigned* rflags = reinterpret_cast<unsigned*>(W0 + ws.flags);
    const int M = B * T;
    const float* mask = (flags & TIP_FWD_KEEP_MASK) ? keep_mask : nullptr;
    if (!mask) keep_scale = 1.f;
    hipError_t e;

#define TIP_TRY(expr, what)                          \
    do {                                             \
        e = (expr);                                  \
        if (e != hipSuccess) return fail_hip(h, e, what); \
    } while (0)

    if (reuse) {
        // the newest row of every window -> slot (frame mod 40) of its stream's ring; while the windows still grow (T < 40) that is all
        // the reuse form does, and the forward below is tip_forward's
        StageScope sc(h, s, "reuse_update");
        TIP_TRY(launch_reuse_update(d, P + L.fused_off, x_imu, x_s, reuse->cache, reuse->frame_ctr, reuse->frame_idx, B, T, s), "reuse_update");
    }
    int plan = reuse_full ? TIP_PLAN_FUSED2 : h->plan;
    if (plan == TIP_PLAN_AUTO) {
        // (a demoted handle — TIP_OPT_DEMOTED, after a lost hand-off — takes no cooperating kernel: the latency plan's GEMV recurrence is one)
        // few streams: the latency plan up to 32 windows (0.17-0.28 ms), then ONE window on FOUR CUs up to #CUs / 4 windows (0.30 ms
        // per step; the latency plan takes 0.36 ms for 40 windows) and on TWO up to #CUs / 2 (0.45 ms against 0.60 for one window
        // per CU) — the window-split encoder, T = 40 only; the latency plan again where that does not apply (<= 64 shorter windows)
        const bool f1s4 = fused2_supported(d, T) && h->f1s_parts != 2 && fused1s_quad_fits(B, cus);
        if (!h->demoted && B <= (f1s4 ? 32 : 48) && latency_supported(d, B, T)) plan = TIP_PLAN_LATENCY;
        else if (!h->demoted && fused2_supported(d, T) && fused1s_fits(B, cus)) plan = TIP_PLAN_FUSED1S;
        else if (!h->demoted && latency_supported(d, B, T)) plan = TIP_PLAN_LATENCY;   // <= 64 streams: spread each window over many CUs
        else plan = fused_supported(d, T) ? TIP_PLAN_FUSED : TIP_PLAN_GENERAL;
    }
    if (plan == TIP_PLAN_FUSED && h->plan == TIP_PLAN_AUTO && !reuse_full) {
        // One window per workgroup with the hybrid row tiling (no hand-offs, 0.527 ms per round of #CUs windows at T = 40), or
        // two windows per workgroup (80 rows = 5 full MFMA row blocks, 1.049 ms per round of 2 x #CUs windows; only the RATIO of
        // the two matters, and #CUs is the stream's effective count): whichever
        // needs less time for this batch.  (The pair-split plan, 0.605 ms per round with 8 hand-offs per pair, lost its
        // place to the hybrid kernel and stays selectable for measurement.)
        const long long cusl = cus;
        const long long rounds_h = (B + cusl - 1) / cusl, rounds_2 = ((B + 1) / 2 + cusl - 1) / cusl;
        plan = (fused2_supported(d, T) && rounds_2 * 1049 < rounds_h * 527) ? TIP_PLAN_FUSED2 : TIP_PLAN_FUSEDH;
    }
    if ((plan == TIP_PLAN_FUSED || plan == TIP_PLAN_FUSEDH) && !fused_supported(d, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (plan == TIP_PLAN_FUSED2 && !fused2_supported(d, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (plan == TIP_PLAN_FUSED1S && !(fused2_supported(d, T) && fused1s_fits(B, cus) && (h->f1s_parts != 4 || fused1s_quad_fits(B, cus))))
        return TIP_ERR_UNSUPPORTED_CONFIG;
    if (plan == TIP_PLAN_LATENCY && !latency_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    float* enc_out = xa;  // encoder output [M, D]
    bool ih_done = false;  // the fused plan also emits the RNN input projection
    bool rnn_done = false;
    bool hall_armed = false;
    int rnn_cluster = h->rnn_cluster;
    if (rnn_cluster == 0 && h->demoted) {
        rnn_cluster = 1;   // one workgroup per 16-window tile: no inter-workgroup hand-off
    } else if (rnn_cluster == 0) {
        // auto: spread one 16-window tile over as many CUs as the tile count leaves idle
        const int ntiles = (B + kRnnTile - 1) / kRnnTile;
        rnn_cluster = 16;
        while (rnn_cluster > 1 && ntiles * rnn_cluster > cus) rnn_cluster >>= 1;
        // rnn_hidden 512: four-row tiles on 4-workgroup clusters at every batch size (76 us at B = 256 against 114 for the best
        // 16-row variant; tools/rnn_variants2.py).  TIP_RNN_ROWS4=0 keeps the 16-row kernels (measurement).
        static const bool rows4 = !(tip_env("TIP_RNN_ROWS4") && tip_env("TIP_RNN_ROWS4")[0] == '0');
        if (rows4 && d.R == 512) rnn_cluster = kRnnRows4;
    }
    bool head_done = false;
    auto arm_hall = [&]() { return rnn_uses_sentinel(d, B, T, rnn_cluster); };
    if (plan == TIP_PLAN_LATENCY) {
        StageScope sc(h, s, "latency_chain");
        const LatencyHead lh{P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, d.S, (flags & TIP_FWD_LAST_ROW_ONLY) != 0, h->flow_epoch, &head_done,
                             reinterpret_cast<unsigned long long*>(W0 + ws.flow)};
        TIP_TRY(launch_latency_plan(d, P + L.fused_off, P + L.whh_frag_off, x_imu, x_s, mask, keep_scale, W0 + ws.lat, hall,
                                    B, T, cus, gd, s, nullptr, cus == h->num_cus && !h->no_flow ? &lh : nullptr), "latency_chain");
        rnn_done = true;
    } else if (plan == TIP_PLAN_FUSED1S) {
        StageScope sc(h, s, "fused_encoder");
        ih_done = true;
        hall_armed = arm_hall();
        TIP_TRY(launch_fused_encoder1s(d, P + L.fused_off, x_imu, x_s, mask, keep_scale, big, hall_armed ? hall : nullptr,
                                       W0 + ws.xchg, B, cus, h->f1s_parts, gd, s), "fused_encoder1s");
    } else if (plan == TIP_PLAN_FUSED2) {
        StageScope sc(h, s, "fused_encoder");
        ih_done = true;
        hall_armed = arm_hall();
        TIP_TRY(launch_fused_encoder2(d, P + L.fused_off, x_imu, x_s, mask, keep_scale, big, hall_armed ? hall : nullptr, B,
                                      cus, s, reuse_full ? reuse->cache : nullptr, reuse_full ? reuse->frame_ctr : nullptr,
                                      reuse_full ? reuse->frame_idx : 0), "fused_encoder2");
    } else if (plan == TIP_PLAN_FUSEDH) {
        StageScope sc(h, s, "fused_encoder");
        ih_done = fused_has_rnn_ih(d);
        hall_armed = ih_done && arm_hall();
        TIP_TRY(launch_fused_encoder_h(d, P + L.fused_off, x_imu, x_s, mask, keep_scale, ih_done ? nullptr : xa,
                                       ih_done ? big : nullptr, hall_armed ? hall : nullptr, B, T, cus, s),
                "fused_encoder_h");
    } else if (plan == TIP_PLAN_FUSED) {
        StageScope sc(h, s, "fused_encoder");
        ih_done = fused_has_rnn_ih(d);
        hall_armed = ih_done && arm_hall();   // the encoder pre-fills its HALL rows
        TIP_TRY(launch_fused_encoder(d, P + L.fused_off, x_imu, x_s, mask, keep_scale, ih_done ? nullptr : xa,
                                     ih_done ? big : nullptr, hall_armed ? hall : nullptr, B, T, cus, s),
                "fused_encoder");
    } else {
        {
            StageScope sc(h, s, "prologue");
            TIP_TRY(launch_prologue(d, x_imu, x_s, mask, keep_scale, big, M, s), "prologue");
        }
        {
            StageScope sc(h, s, "in_linear");
            TIP_TRY(launch_gemm(big, d.InPad, P + L.in_lin.w_off, L.in_lin.Kpad, P + L.in_lin.b_off, nullptr, 0, xa,
                                d.D, M, d.D, L.in_lin.Npad, 0, s), "in_linear");
        }
        for (int l = 0; l < d.L; ++l) {
            const PackedLayer& pl = L.layers[l];
            {
                StageScope sc(h, s, "qkv_gemm");
                TIP_TRY(linear_gemm(P, pl.qkv, xa, d.D, nullptr, 0, big, 3 * d.D, M, 0, s), "qkv_gemm");
            }
            {
                StageScope sc(h, s, "attention");
                TIP_TRY(launch_attention(d, big, att, B, T, s), "attention");
            }
            {
                StageScope sc(h, s, "out_proj_gemm");
                TIP_TRY(linear_gemm(P, pl.out, att, d.D, xa, d.D, xb, d.D, M, 2, s), "out_proj_gemm");
            }
            {
                StageScope sc(h, s, "layernorm1");
                TIP_TRY(launch_layernorm(xb, P + pl.g1_off, P + pl.be1_off, M, d.D, s), "layernorm1");
            }
            {
                StageScope sc(h, s, "ffn1_gemm");
                TIP_TRY(linear_gemm(P, pl.ff1, xb, d.D, nullptr, 0, big, d.F, M, 1, s), "ffn1_gemm");
            }
            {
                StageScope sc(h, s, "ffn2_gemm");
                TIP_TRY(linear_gemm(P, pl.ff2, big, d.F, xb, d.D, xa, d.D, M, 2, s), "ffn2_gemm");
            }
            {
                StageScope sc(h, s, "layernorm2");
                TIP_TRY(launch_layernorm(xa, P + pl.g2_off, P + pl.be2_off, M, d.D, s), "layernorm2");
            }
        }
    }

    const bool last_only = (flags & TIP_FWD_LAST_ROW_ONLY) != 0;
    const float* head_in = enc_out;
    int head_ld = d.D;
    if (d.with_rnn && rnn_done) {
        head_in = hall;
        head_ld = d.R;
    } else if (d.with_rnn) {
        if (!ih_done) {
            StageScope sc(h, s, "rnn_ih_gemm");
            TIP_TRY(launch_gemm(enc_out, d.D, P + L.rnn_ih.w_off, L.rnn_ih.Kpad, P + L.rnn_ih.b_off, nullptr, 0, big,
                                d.R, M, d.R, L.rnn_ih.Npad, 0, s), "rnn_ih_gemm");
        }
        {
            StageScope sc(h, s, "rnn_recurrence");
            TIP_TRY(launch_rnn(d, big, P + L.whh_frag_off, hall, rflags, B, T, rnn_cluster, cus, hall_armed, gd, s), "rnn_recurrence");
        }
        head_in = hall;
        head_ld = d.R;
    }
    if (!head_done) {
        StageScope sc(h, s, "out_linear");
        const int Kh = d.with_rnn ? d.R : d.D;
        // rows the projection runs on: all M, or row T-1 of every window (real_time_runner_minimal.py:150)
        const float* hA = last_only ? head_in + (size_t)(T - 1) * head_ld : head_in;
        const long long hlda = last_only ? (long long)T * head_ld : head_ld;
        const int hM = last_only ? B : M;
        if (plan == TIP_PLAN_LATENCY) {
            TIP_TRY(launch_latency_head(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, hM, d.S, s), "out_linear");
        } else {
            // window lengths that are multiples of 40 (the paper's and the scaled configuration's): the register-resident
            // kernel for both forms of the output (bit-identical last rows); everything else: head_gemm_kernel for both
            static const bool ksplit = !(tip_env("TIP_HEAD") && tip_env("TIP_HEAD")[0] == 'o');   // TIP_HEAD=old: measurement
            hipError_t he = hipErrorInvalidValue;
            if (ksplit && T % 40 == 0)
                he = launch_head_ksplit(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, hM, d.S, Kh, last_only, cus, s);
            if (he == hipErrorInvalidValue)
                he = launch_head_gemm(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, hM, d.S, Kh, s);
            TIP_TRY(he, "out_linear");
        }
    }
#undef TIP_TRY
    h->forward_count++;
    return TIP_OK;
}

int tip_draw_keep_mask(float p_state, unsigned long long state_seed, float* mask, size_t n, tip_stream_t stream) {
    unsigned key = 0, thresh = 0;
    if ((!mask && n) || !state_mask_params(p_state, state_seed, &key, &thresh)) return TIP_ERR_INVALID_ARG;
    return launch_keep_mask(mask, n, key, thresh, static_cast<hipStream_t>(stream)) == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

int tip_forward_dropout(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                        const float* keep_mask, float keep_scale, float p_state, unsigned long long state_seed, float p_drop,
                        unsigned long long seed, void* workspace, size_t workspace_bytes, tip_stream_t stream) {
    if (!h || !x_imu || !x_s || !y || B < 0 || T < 1) return TIP_ERR_INVALID_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return TIP_ERR_INVALID_ARG;
    unsigned mkey = 0, mthresh = 0;
    if (!(flags & TIP_FWD_KEEP_MASK) && p_state > 0.f && !state_mask_params(p_state, state_seed, &mkey, &mthresh)) return TIP_ERR_INVALID_ARG;
    if ((flags & TIP_FWD_KEEP_MASK) && !keep_mask) return TIP_ERR_INVALID_ARG;
    if (!h->packed_dev) return TIP_ERR_NOT_READY;
    if (tip_check(h, 0) != TIP_OK) return TIP_ERR_HANDOFF;
    if (B == 0) return TIP_OK;
    const Dims& d = h->d;
    // the few-stream latency plan is the only one with the dropout sites; a demoted handle takes no cooperating kernel (its GEMV
    // recurrence is one): the caller falls back to tip_train_forward
    if (h->demoted || !latency_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    const Workspace ws = carve_workspace(d, B, T);
    if (!workspace || reinterpret_cast<uintptr_t>(workspace) % 256 || workspace_bytes < ws.total_bytes) return TIP_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int cus = effective_cus(h->num_cus, s);
    CoopSerial serial(h->device, s);
    if (serial.status != hipSuccess) return fail_hip(h, serial.status, "stream serialisation");
    const float* P = h->packed_dev;
    const PackedLayout& L = h->lay;
    float* W0 = static_cast<float*>(workspace);
    float* hall = W0 + ws.hall;
    const float* mask = (flags & TIP_FWD_KEEP_MASK) ? keep_mask : nullptr;
    if (!mask && !mthresh) keep_scale = 1.f;
    TrainDropout td = make_train_dropout(p_drop, seed);
    td.mkey = mkey;
    td.mthresh = mthresh;
    hipError_t e;
    bool head_done = false;
    {
        StageScope sc(h, s, "latency_chain");
        const LatencyHead lh{P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, d.S, (flags & TIP_FWD_LAST_ROW_ONLY) != 0, h->flow_epoch, &head_done,
                             reinterpret_cast<unsigned long long*>(W0 + ws.flow)};
        e = launch_latency_plan(d, P + L.fused_off, P + L.whh_frag_off, x_imu, x_s, mask, keep_scale, W0 + ws.lat, hall, B, T, cus,
                                h->guard(), s, &td, cus == h->num_cus && !h->no_flow ? &lh : nullptr);
        if (e != hipSuccess) return fail_hip(h, e, "latency_chain");
    }
    if (!head_done) {
        StageScope sc(h, s, "out_linear");
        const bool last_only = (flags & TIP_FWD_LAST_ROW_ONLY) != 0;
        const float* hA = last_only ? hall + (size_t)(T - 1) * d.R : hall;
        const long long hlda = last_only ? (long long)T * d.R : d.R;
        e = launch_latency_head(hA, hlda, P + L.out_frag_off, P + L.out_lin.b_off, y, d.S, last_only ? B : B * T, d.S, s);
        if (e != hipSuccess) return fail_hip(h, e, "out_linear");
    }
    h->forward_count++;
    return TIP_OK;
}

}  // extern "C"
static_assert(tip::kRnnRows4 == TIP_RNN_CLUSTER_ROWS4, "option value");
