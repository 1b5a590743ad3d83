// tip_rnnh.hip — tanh recurrence (simple_transformer_with_state.py:98-99) WITH the output projection (:102) inside its hop wait.
//
// rnn_rows4_kernel (tip_general.hip) spends 1 000 of a step's 4 500 cycles waiting for its partners' state slices to cross the
// XCD's L2 with the matrix pipe idle, and the projection y = h W_out^T + b then runs as a launch of its own that re-reads every
// state row from HBM (17 us + a kernel boundary at B = 256).  Here the same four-workgroup cluster computes y_{t-1} between
// the moment it has stored its slice of h_t and the moment its partners' slices have arrived:
//   * tile = FOUR windows, cluster = FOUR workgroups of one XCD, member `cid` owns state columns 128 cid .. + 127, each of its
//     8 waves 16 columns x all 512 k in 128 VGPRs — the recurrence is rnn_rows4_kernel's, instruction for instruction (same
//     fragments, same four k-chains per lane group, same reduction): h is bit-identical to that kernel's;
//   * every member holds the whole h_{t-1} tile in LDS (it pulled it for the recurrence), so the projection needs no further
//     exchange: wave w of member cid owns output columns 4 q .. 4 q + 3, q = 8 cid + w, with its 4 x 512 slice of W_out in 32
//     VGPRs, and multiplies with the same 4x4x1 MFMA laid out as 16 k-groups x ONE column quad (32 MFMAs per wave and step,
//     a quarter of the recurrence's); columns 128 .. 130 (quad 32) are k-split over the eight waves of member 3 and meet in LDS;
//   * the polls for h_t are issued INSIDE that MFMA stream, so that their L2 round trips run under it;
//   * nothing of the state goes to HBM any more: the hand-off travels through a two-slot ring of 8-byte {value, tag} granules
//     per cluster (32 KB, L2 resident), tag = (tile sequence, step) — no 21-MB sentinel fill of HALL by the encoder, no 21-MB
//     state write, no 21-MB re-read by the projection.  The ring is armed (all ones, which no tag equals) once per launch: by
//     the encoder that runs in front (a 16-byte store per thread), else by a memset.
// Hand-off protocol otherwise as rnn_rows4_kernel: same-XCD placement verified through launch-tagged exchange words (plain
// stores + L1-bypassing loads when true, write-through stores otherwise), every wait bounded, a wait that gives up poisons what
// it feeds with NaN and raises the sticky TIP_ERR_HANDOFF; a member dropped by TIP_OPT_FAULT_INJECT still writes its output
// columns (as NaN), never leaves them stale.
// Numerics of y: per output, k is summed in four chains per k-group (s = 0..3 of every 64-k block, ascending), (c0 + c1) +
// (c2 + c3), then the 16 k-groups pairwise: g with g + 8, + 4, + 2, + 1 — for every batch size and both output forms
// (row T-1 of the full output == the last-row-only output, bit for bit).
#include "tip_internal.h"
#include "tip_layernorm.h"
#include "tip_rnnh.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ unsigned g_spin_timeouts_rnnh;
__device__ __forceinline__ void rnnh_note_timeout(unsigned* err) {
    atomicAdd(&g_spin_timeouts_rnnh, 1u);
    guard_report(err);
}
hipError_t read_spin_timeouts_rnnh(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spin_timeouts_rnnh), sizeof(unsigned));
}

// measurement only (TIP_RNNH_TRACE=1): s_memtime stamps of workgroup 0 / thread 0, 12 slots per step (tools/rnnh_trace.py)
__device__ unsigned long long g_rnnh_trace[12 * 64];
#define RH_STAMP(slot) do { if (TRACE && (a.knob & 256) && (!(a.knob & 512) || (slot) == 2 || (slot) == 7 || (slot) == 0) && blockIdx.x == 0 && tid == 0 && t < 64) g_rnnh_trace[t * 12 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RH_NOTE(slot, val) do { if (TRACE && (a.knob & 256) && blockIdx.x == 0 && tid == 0 && t < 64) g_rnnh_trace[t * 12 + (slot)] = (val); } while (0)

namespace rh {
constexpr int R = 512, KB = R / 16, LD = R + 16, ROWS = 4, CLUSTER = 4, THREADS = 512;
constexpr int HBUF_FLOATS = 2 * ROWS * LD;          // two h tiles
constexpr int PART_FLOATS = 2 * 8 * 16;             // quad 32: [buffer][wave][row * 4 + column]
constexpr int LDS_BYTES = (HBUF_FLOATS + PART_FLOATS) * 4;
constexpr int SLOT_BYTES = ROWS * R * 8;            // one ring slot: 4 rows x 512 {value, tag}
constexpr int RING_BYTES = 2 * SLOT_BYTES;          // per cluster
}  // namespace rh

constexpr int kMaxGroups = 72;   // clusters in flight the ring is sized for (a 288-CU stream); the launcher never uses more
size_t rnnh_ring_bytes(int B) {
    const int tiles = (B + rh::ROWS - 1) / rh::ROWS;
    return (size_t)(tiles < kMaxGroups ? tiles : kMaxGroups) * rh::RING_BYTES;
}

struct RnnHeadArgs {
    const float* ih;        // [B][T][512] input projection incl. b_ih + b_hh
    const float* whh_frag;  // W_hh in B-fragment order [32 nb][32 kb][64][4]
    const float* wout;      // W_out row-major [>= 132 rows, zero padded][512]
    const float* bout;      // [>= 132]
    float* y;               // [B][T][S] or [B][S]
    unsigned* ring;         // ngroups x RING_BYTES, armed with all ones
    unsigned* flags;        // XCC-exchange words [ngroups][4]
    int B, T, ntiles, ngroups, ih_bytes, y_bytes, S, last_only;
    unsigned etag;
    int knob;               // poll placement (measurement while tuning)
    Guard gd;
};

template <bool TRACE>
__global__ __launch_bounds__(rh::THREADS) void rnn_head_kernel(RnnHeadArgs a) {
    using namespace rh;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* part = smem + HBUF_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int T = a.T;
    // cluster membership: members 8 workgroup ids apart (one XCD as workgroups are observed to be dealt; verified below)
    const int cid = (blockIdx.x >> 3) % CLUSTER, group = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / CLUSTER);
    if (group >= a.ngroups) return;
    const int nb = cid * 8 + wave;                                  // global 16-column block of the state / column quad of y
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ih), 0, a.ih_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.ring) + (size_t)group * RING_BYTES, 0, RING_BYTES, 0x00020000);
    const bool dead = (a.gd.fault & 2) && group == 0 && cid == 1;   // TIP_OPT_FAULT_INJECT: this member never publishes a state
    const unsigned spin_big = guard_spin_limit(a.gd.fault, 1u << 22), spin_pull = guard_spin_limit(a.gd.fault, 1u << 20);

    // step 0 needs no W_hh: its input term is requested in front of the 256-KB weight load (vector memory returns in order)
    const int tpg = (a.ntiles + a.ngroups - 1) / a.ngroups;
    const unsigned rowbytes = (unsigned)T * R * 4;
    const int col_out = nb * 16 + l15;
    auto ih_voff = [&](int tile) { return (int)((unsigned)(tile * ROWS + lg) * rowbytes + (unsigned)col_out * 4u); };
    float ihn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, ih_voff(group), 0, 0));
    float4 wreg[KB];
    {
        const float4* wf = reinterpret_cast<const float4*>(a.whh_frag) + (size_t)nb * KB * 64 + lane;
#pragma unroll
        for (int k = 0; k < KB; ++k) wreg[k] = wf[(size_t)k * 64];
    }
    // W_out slice of this wave's column quad: lane (g = lane >> 2, c = lane & 3) holds W_out[4 q + c][64 J + 4 g + s], J < 8
    float4 wout[8];
    {
        const float* wp = a.wout + (size_t)(4 * nb + (lane & 3)) * R + 4 * (lane >> 2);
#pragma unroll
        for (int J = 0; J < 8; ++J) wout[J] = *reinterpret_cast<const float4*>(wp + 64 * J);
    }
    // ... and of quad 32 (columns 128..131), k-block `wave`: member 3 only
    const float4 wq32 = *reinterpret_cast<const float4*>(a.wout + (size_t)(128 + (lane & 3)) * R + 64 * wave + 4 * (lane >> 2));
    const int ycol = 4 * nb + (lane & 3);
    const float ybias = a.bout[ycol];
    const float ybias32 = a.bout[128 + (lane & 3)];

    // same-XCD fast path, verified at run time through launch-tagged exchange words (as rnn_rows4_kernel)
    __shared__ int s_same_xcd;
    if (tid == 0 && dead) s_same_xcd = 0;
    if (tid == 0 && !dead) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        const unsigned mine = (a.etag << 5) | (xcc + 1u);
        __hip_atomic_store(a.flags + group * CLUSTER + cid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int m = 0; m < CLUSTER; ++m) {
            unsigned v = 0;
            bool here = false;
            for (unsigned spins = 0; spins < spin_big; ++spins) {
                v = __hip_atomic_load(a.flags + group * CLUSTER + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                here = (v >> 5) == a.etag && (v & 31u) != 0u;
                if (here) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!here) rnnh_note_timeout(a.gd.err);
            same &= (v == mine);
        }
        s_same_xcd = (same && !(a.gd.fault & 8)) ? 1 : 0;
    }
    __syncthreads();
    const bool same_xcd = s_same_xcd != 0;

    const int aoff = (lane & 3) * LD + lg * 4;                      // recurrence A operand: row lane & 3, k = 16 kb + 4 lg ..
    const int aoffh = (lane & 3) * LD + 4 * (lane >> 2);             // projection A operand: row lane & 3, k = 64 J + 4 g ..
    const int prow = tid >> 7, pcol = (tid & 127) * 4;               // this thread's piece of a pulled tile: 4 columns of one row
    const int vpull = (prow * R + pcol) * 8;
    const int lds_w = prow * LD + pcol;
    const int vout = (lg * R + col_out) * 8;
    bool poisoned = dead;                                            // (wave-uniform)
    unsigned n_miss = 0, n_rounds = 0;                               // TRACE: steps whose in-stream poll missed / extra poll rounds (this wave)

    // a poll = this thread's 32 bytes of slot `sl` (four granules), L1-bypassing loads
    auto poll_issue = [&](u32x4 (&v)[2], int sl) {
        asm volatile("" ::: "memory");   // (loop-invariant addresses: without this the optimiser polls a register)
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vpull, sl * SLOT_BYTES, 16);
        v[1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, vpull + 16, sl * SLOT_BYTES, 16);
    };
    auto poll_pending = [&](const u32x4 (&v)[2], unsigned tag) -> bool {
        const bool pend = v[0].y != tag || v[0].w != tag || v[1].y != tag || v[1].w != tag;
        return __builtin_amdgcn_ballot_w64(pend) != 0;
    };

    // reduce a projection accumulator over the 16 k-groups (lane >> 2): g with g + 8 (rows 0, 2 stay / go), g + 4 (rows 1, 3), then
    // + 2, + 1 inside the 16-lane row; lane 16 row + c of every row ends with (row, column c)
    auto kreduce = [&](const f32x4& pv) -> float {
        float b0 = pv[0], b2 = pv[2];
        swap32(b0, b2);
        float m0 = b0 + b2;
        float b1 = pv[1], b3 = pv[3];
        swap32(b1, b3);
        float m1 = b1 + b3;
        swap16(m0, m1);
        float m = m0 + m1;
        m += dpp_peer<0x128>(m);   // row_ror:8
        m += dpp_peer<0x124>(m);   // row_ror:4
        return m;
    };
    const bool ylane = (lane & 12) == 0;
    const unsigned ycol_b = ycol < a.S ? (unsigned)ycol * 4u : 0x80000000u;                       // (bad column: out of the descriptor's range)
    const unsigned ycol32_b = 128 + (lane & 3) < a.S ? (unsigned)(128 + (lane & 3)) * 4u : 0x80000000u;
    const unsigned yrow_b = a.last_only ? (unsigned)a.S * 4u : (unsigned)T * (unsigned)a.S * 4u;
    const unsigned frame_stride = a.last_only ? 0u : (unsigned)a.S * 4u;

    for (int q0 = 0; q0 < tpg; ++q0) {
        const int tile = group + a.ngroups * q0;
        if (tile >= a.ntiles) break;                                 // (the whole cluster agrees)
        const unsigned tagbase = ((unsigned)q0 << 8) + 1u;
        // ring slots alternate with a parity that runs on ACROSS the tiles of a cluster: the slot a member writes step 0 of the next
        // tile into is the one whose content (step T - 2 of this tile) every partner has consumed — with a per-tile parity and an
        // odd T it would be the slot of step T - 1, which a slower partner may not have read yet
        const int sbase = (q0 * T) & 1;
        const int ihv_off = ih_voff(tile);
        if (q0 > 0) ihn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, ihv_off, 0, 0));
        const unsigned ywin = (unsigned)(tile * ROWS + (lane >> 4)) * yrow_b;   // window tile * 4 + row; the frame is added per step

        // raw projection of the tile in `hb` (y_{tp} before the reduction over the k-groups): 32 MFMAs; `polls`: the poll for the NEXT
        // state is issued inside the MFMA stream (slot `sl`).  Nothing else sits in the hop wait: the reduction, the store and
        // quad 32 ride under the NEXT step's recurrence MFMAs (finish_y, quad32_partial), where the VALU has idle issue slots.
        auto project_raw = [&](const float* hb, bool polls, int sl, u32x4 (&q1)[2]) -> f32x4 {
            const float* hp = hb + aoffh;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0, d2 = d0, d3 = d0;
            float4 ha[4];   // A fragments four k-blocks ahead of their MFMAs
#pragma unroll
            for (int J = 0; J < 4; ++J) ha[J] = *reinterpret_cast<const float4*>(hp + 64 * J);
#pragma unroll
            for (int J = 0; J < 8; ++J) {
                const float4 hk = ha[J & 3];
                if (J + 4 < 8) ha[J & 3] = *reinterpret_cast<const float4*>(hp + 64 * (J + 4));
                d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hk.x, wout[J].x, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hk.y, wout[J].y, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(hk.z, wout[J].z, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_4x4x1f32(hk.w, wout[J].w, d3, 0, 0, 0);
                if (J == (a.knob & 7)) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (polls) poll_issue(q1, sl);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            return (d0 + d1) + (d2 + d3);
        };
        // quad 32 (columns 128..130) of y_{tp} from the tile in `hb`: member 3's wave w takes k-block w; the eight partials meet in LDS
        auto quad32_partial = [&](const float* hb, int tp) {
            if (cid == 3) {
                const float4 hq = *reinterpret_cast<const float4*>(hb + aoffh + 64 * wave);
                f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0, e2 = e0, e3 = e0;
                e0 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq.x, wq32.x, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq.y, wq32.y, e1, 0, 0, 0);
                e2 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq.z, wq32.z, e2, 0, 0, 0);
                e3 = __builtin_amdgcn_mfma_f32_4x4x1f32(hq.w, wq32.w, e3, 0, 0, 0);
                const float pv = kreduce((e0 + e1) + (e2 + e3));
                if (ylane) part[(tp & 1) * 128 + wave * 16 + (lane >> 4) * 4 + (lane & 3)] = pv;
            }
        };
        auto finish_y = [&](const f32x4& praw, int tp) {
            const float yv = kreduce(praw) + ybias;
            if (ylane) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(yv), yrs, (int)(ywin + (unsigned)tp * frame_stride + ycol_b), 0, 0);
        };
        // quad 32 of y_{tp}: the eight k-partials of member 3's waves, in a fixed order (after a barrier behind project())
        auto quad32_store = [&](int tp) {
            if (cid == 3 && wave == 0 && ylane) {
                const float* pp = part + (tp & 1) * 128 + (lane >> 4) * 4 + (lane & 3);
                const float sv = ((pp[0] + pp[16]) + (pp[32] + pp[48])) + ((pp[64] + pp[80]) + (pp[96] + pp[112]));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sv + ybias32), yrs, (int)(ywin + (unsigned)tp * frame_stride + ycol32_b), 0, 0);
            }
        };

        // One step as a straight-line body per (FIRST, PROJ): with the two forms of the hop wait — projection with the poll inside, or the
        // poll alone — as branches of ONE body, the poll's registers are "in flight" on both paths at the join and the projection's first
        // register write waits for vmcnt(0), i.e. for the acknowledgement of the h store just issued (seen in the ISA: ~500 cycles a step).
        f32x4 praw = {0.f, 0.f, 0.f, 0.f};   // raw projection of the previous step's hop wait (y_{t-2} on entry of step t)
        auto step = [&](auto FIRST_C, auto PROJ_C, int t) {
            constexpr bool FIRST = decltype(FIRST_C)::value, PROJ = decltype(PROJ_C)::value;
            // on entry: h_{t-1} (t > 0) is in LDS buffer t & 1, barrier passed
            float* buf = smem + (t & 1) * (ROWS * LD);
            RH_STAMP(0);
            if (a.knob & 1024) __builtin_amdgcn_s_setprio(2);
            // The input term of THIS step, requested here and used behind the MFMAs (2 300 cycles on).  Not a step ahead: a loaded value
            // carried around the loop makes the back-edge a vmcnt(0) — over the acknowledgement of the y stores issued just before it.
            const float ihv = FIRST ? ihn : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, ihv_off, t * (R * 4), 0));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
            if (!FIRST) {
                // A fragments four k-blocks ahead of their MFMAs
                const float* ap = buf + aoff;
                float4 af[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) af[k] = *reinterpret_cast<const float4*>(ap + k * 16);
#pragma unroll
                for (int k = 0; k < KB; ++k) {
                    const float4 ak = af[k & 3];
                    if (k + 4 < KB) af[k & 3] = *reinterpret_cast<const float4*>(ap + (k + 4) * 16);
                    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.x, wreg[k].x, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.y, wreg[k].y, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.z, wreg[k].z, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.w, wreg[k].w, c3, 0, 0, 0);
                    if (PROJ && k == 3) {    // under the MFMA stream: finish y_{t-2} (reduction, bias, store) ...
                        __builtin_amdgcn_sched_barrier(0);
                        if (t >= 2) finish_y(praw, t - 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (PROJ && k == 12) {   // ... and quad 32 of y_{t-1} from the tile this step reads anyway
                        __builtin_amdgcn_sched_barrier(0);
                        quad32_partial(buf, t - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            const f32x4 p = (c0 + c1) + (c2 + c3);   // registers = rows 0..3, one k-partial per lane group
            float a0 = p[0], a2 = p[2];
            swap32(a0, a2);
            float k0 = a0 + a2;
            float a1 = p[1], a3 = p[3];
            swap32(a1, a3);
            float k1 = a1 + a3;
            swap16(k0, k1);
            const float acc = k0 + k1;               // lane (lg, l15): row lg, column l15 of this wave's block
            if (TRACE && (a.knob & 256)) { asm volatile("" :: "v"(acc)); RH_STAMP(1); }
            float hv = tip_tanh(acc + ihv);
            if (hv != hv) hv = __uint_as_float(kPoisonBits);
            const int sl = (sbase + t) & 1;
            const u32x2 gr = {__float_as_uint(hv), tagbase + (unsigned)t};
            if (!dead) {
                // same XCD (verified): a plain store lands in the shared L2 (L1 is write-through); otherwise sc1 = write-through
                if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(gr, xrs, vout, sl * SLOT_BYTES, 0);
                else __builtin_amdgcn_raw_buffer_store_b64(gr, xrs, vout, sl * SLOT_BYTES, 16);
            }
            RH_STAMP(2);
            // the hop wait's matrix work yields to a co-resident wave that is still in its recurrence phase: what it delays there
            // is the slowest wave's state store, i.e. the whole cluster's next step
            if (a.knob & 1024) __builtin_amdgcn_s_setprio(0);
            // ---- hop wait: y_{t-1} from the tile in LDS, the polls for h_t inside its MFMA stream ----
            u32x4 p1[2];
            if (PROJ) praw = project_raw(buf, true, sl, p1);
            else poll_issue(p1, sl);
            // the granule's registers stay live up to here: reused any earlier, the compiler orders the reuse behind the STORE (a wait for
            // its acknowledgement, vmcnt(0), in front of the projection's first LDS read: ~500 cycles of every step)
            asm volatile("" :: "v"(gr.x), "v"(gr.y));
            RH_STAMP(4);
            // ---- h_t: the poll in flight, then as many more as it takes ----
            {
                const unsigned tag = tagbase + (unsigned)t;
                u32x4 v[2] = {p1[0], p1[1]};
                bool pend = poll_pending(v, tag);   // (always: a path on which the poll is never waited for leaves its registers "in flight"
                pend = pend || poisoned;            //  for the compiler, and the next write to them becomes a vmcnt(0))
                RH_STAMP(5);
                RH_NOTE(9, pend ? 1ull : 0ull);
                if (TRACE && pend) ++n_miss;
                if (pend) {
                    bool gave_up = true;
                    const unsigned lim = poisoned ? 1u : spin_pull;
                    for (unsigned spins = 0; spins < lim; ++spins) {
                        poll_issue(v, sl);
                        if (TRACE) ++n_rounds;
                        RH_NOTE(11, (unsigned long long)spins + 1ull);
                        if (!poll_pending(v, tag)) { gave_up = false; break; }
                        if (!same_xcd) __builtin_amdgcn_s_sleep(2);   // cross-XCD polls travel the fabric: pace them
                    }
                    if (gave_up) {   // (wave-uniform) what never arrived becomes NaN
                        if (!poisoned && lane == 0) rnnh_note_timeout(a.gd.err);
                        poisoned = true;
                        if (v[0].y != tag) v[0].x = kPoisonBits;
                        if (v[0].w != tag) v[0].z = kPoisonBits;
                        if (v[1].y != tag) v[1].x = kPoisonBits;
                        if (v[1].w != tag) v[1].z = kPoisonBits;
                    }
                }
                RH_STAMP(7);
                *reinterpret_cast<u32x4*>(smem + ((t + 1) & 1) * (ROWS * LD) + lds_w) = (u32x4){v[0].x, v[0].z, v[1].x, v[1].z};
                if (TRACE && (a.knob & 256)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); RH_STAMP(8); }
                __syncthreads();   // the only barrier of a step (the tile buffers alternate)
            }
            if (PROJ) quad32_store(t - 1);
        };
        using std::integral_constant;
        step(integral_constant<bool, true>{}, integral_constant<bool, false>{}, 0);
        if (a.last_only) {
#pragma unroll 1
            for (int t = 1; t < T; ++t) step(integral_constant<bool, false>{}, integral_constant<bool, false>{}, t);
        } else {
#pragma unroll 1
            for (int t = 1; t < T; ++t) step(integral_constant<bool, false>{}, integral_constant<bool, true>{}, t);
        }
        // y_{T-2} (its raw projection is still in registers) and the last state's projection (the only one of a last-row-only forward)
        {
            if (!a.last_only && T >= 2) finish_y(praw, T - 2);
            u32x4 p1[2];
            const float* hb = smem + (T & 1) * (ROWS * LD);
            quad32_partial(hb, T - 1);
            finish_y(project_raw(hb, false, 0, p1), T - 1);
            __syncthreads();
            quad32_store(T - 1);
            __syncthreads();   // the LDS buffers are free for the next tile
        }
    }
    if (TRACE && lane == 0) {   // [12 * 63 + ..]: waves, steps with a missed in-stream poll, extra poll rounds (all workgroups)
        atomicAdd(&g_rnnh_trace[12 * 63 + 0], 1ull);
        atomicAdd(&g_rnnh_trace[12 * 63 + 1], (unsigned long long)n_miss);
        atomicAdd(&g_rnnh_trace[12 * 63 + 2], (unsigned long long)n_rounds);
    }
    // leave the XCC-exchange word cleared for a replay of this launch from a HIP graph (same tag): see rnn_resident_kernel
    if (T >= 2 && tid == 0 && !dead) __hip_atomic_store(a.flags + group * CLUSTER + cid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

bool rnnh_supported(const Dims& d, int T) { return d.with_rnn && d.R == 512 && d.S > 128 && d.S <= 132 && T >= 1 && T <= 250; }

hipError_t launch_rnn_head(const Dims& d, const float* ih, const float* whh_frag, const float* wout, const float* bout, float* y,
                           float* ring, unsigned* flags, int B, int T, bool last_only, bool ring_armed, int num_cus, unsigned etag,
                           const Guard& gd, hipStream_t s) {
    using namespace rh;
    if (!rnnh_supported(d, T)) return hipErrorInvalidValue;
    const int ntiles = (B + ROWS - 1) / ROWS;
    int groups = ntiles;
    if (groups * CLUSTER > num_cus) groups = num_cus / CLUSTER;
    if (groups > kMaxGroups) groups = kMaxGroups;
    if (groups < 1) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_head_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // every member must be resident while its partners wait for it
    hipError_t ce = check_coresident(rnn_head_kernel<false>, THREADS, LDS_BYTES, groups * CLUSTER, num_cus, &occ);
    if (ce != hipSuccess) return ce;
    if (!ring_armed) {
        hipError_t e = hipMemsetAsync(ring, 0xFF, (size_t)groups * RING_BYTES, s);
        if (e != hipSuccess) return e;
    }
    RnnHeadArgs a;
    a.ih = ih; a.whh_frag = whh_frag; a.wout = wout; a.bout = bout; a.y = y;
    a.ring = reinterpret_cast<unsigned*>(ring); a.flags = flags;
    a.B = B; a.T = T; a.ntiles = ntiles; a.ngroups = groups;
    a.ih_bytes = (int)((long long)B * T * R * 4);
    a.y_bytes = (int)((long long)B * (last_only ? 1 : T) * d.S * 4);
    a.S = d.S; a.last_only = last_only ? 1 : 0; a.etag = etag; a.gd = gd;
    static int knob = -1;
    if (knob < 0) knob = getenv("TIP_RNNH_KNOB") ? atoi(getenv("TIP_RNNH_KNOB")) : 5;
    a.knob = knob;
    // members of a cluster are taken 8 workgroup ids apart (one XCD): whole rounds of 8 clusters
    const dim3 grid((groups + 7) / 8 * 8 * CLUSTER), block(THREADS);
    static int trace = -1;
    if (trace < 0) trace = getenv("TIP_RNNH_TRACE") ? 1 : 0;
    if (trace) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rnn_head_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rnn_head_kernel<true>), grid, block, LDS_BYTES, s, a);
    } else {
        hipLaunchKernelGGL((rnn_head_kernel<false>), grid, block, LDS_BYTES, s, a);
    }
    return hipGetLastError();
}

}  // namespace tip
extern "C" int tip_debug_read_rnnh_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 12 * 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_rnnh_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
