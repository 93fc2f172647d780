// rp_engine.hip -- MI355X (gfx950 / CDNA4) self-play bin-packing engine: game rules, the
// transposition-DAG Monte-Carlo tree search and the evaluator plumbing, behind the C ABI of
// include/rp_engine.h.
//
// Execution model: one 64-lane wavefront owns one game slot.  A grid row lives in the lane with the
// same index (rows are W-bit masks, W <= 64, H <= 64), so the reference's per-cell Python loops
// (BinPackingLogic.py:47-109) become popcounts, ballots and readlanes over a register-resident
// board.  Every tree of a slot is a structure of arrays in HBM: 32-byte node headers, bit-packed
// node keys (the state), per-node CONTIGUOUS runs of the legal moves' (action u16, prior f32) -- 6 bytes
// per legal move, written once at expansion -- and per-node blocks of VISITED edges only
// (32-byte records: Q, P, N, child, index; grown by doubling), because a search visits ~2 % of the edges it creates.
// A wave reads a node's priors and visited records as coalesced 64-lane loads; a node with more legal moves than lanes keeps its best
// unvisited move in the header, so a selection scores the visited records plus that one candidate.  Transpositions (the reference keys
// its dicts by the full state, MCTS_bpp.py:20-26,76) go through a per-slot open-addressing table probed one aligned 16-slot bucket (one
// 128-byte line) at a time.  A slot's regions lie back to back in ONE slab (DP::slab_stride), and everything that is the same in all lanes of a wave -- the slot
// id, region pointers, node headers, counters -- is kept in scalar registers (wave_in_block(), uni()).
//
// Numerics: compiled with -ffp-contract=off; the PUCT score is float64, Q follows the NumPy promotion
// state machine of the reference's backup expression, the prior is renormalised in NumPy's pairwise
// summation order.  See DESIGN.md.
//
// Reference paths cited below are relative to /root/reference/xw_mcts.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/rp_engine.h"

#pragma clang fp contract(off)

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

#define NONE32 0xFFFFFFFFu
#define NSA_MASK 0x3FFFFFFFu
#define WAVES_PER_BLOCK 4
#define MAX_LEAVES 128
#define MAX_MASK_WORDS 256 /* A <= 8192 */

enum { CNT_SIMS = 0, CNT_EXPAND, CNT_TERMINAL, CNT_PATH, CNT_NVALID_SEL, CNT_NVALID_LEAF, CNT_TRANSPOSE, CNT_NODES,
       CNT_MOVES, CNT_EPISODES, CNT_PROBES, CNT_KEYBYTES, CNT_VIS_SEL, CNT_VIS_NEW, CNT_RES0, CNT_RES1, CNT_N };
enum { ERR_NODE_CAP = 1, ERR_EDGE_CAP = 2, ERR_TABLE_FULL = 3, ERR_BAD_ACTION = 4, ERR_PATH = 5, ERR_FINISHED_CAP = 6, ERR_EXAMPLES_CAP = 7, ERR_VIS_CAP = 8, ERR_BAD_EXAMPLE = 9 };

struct NodeHdr {    // 32 bytes = two dwordx4 loads per visited node
    u32 ns;         // Ns[s]          (MCTS_bpp.py:103,138)
    u32 prior_off;  // first entry of this node's legal-move run in pAct / pPi
    u32 vis_off;    // first entry of its visited-edge block
    u16 n_valid;    // number of legal moves (Vs[s], :102)
    u16 vis_n;      // visited edges = entries in use
    u16 vis_cap;    // entries allocated (0 until the node is first selected at)
    int8_t term;    // Es[s]: 0 not ended, +1 / -1 ranked outcome (:78-83)
    u8 flags;       // bit 0 expanded (s in Ps), bit 1 uniform-fallback prior (:93-100), bits 2-3 kind of term
    u8 depth;       // items already placed in this state = level of the node in the game
    u8 pad;
    u16 best_k;     // the best UNVISITED legal move (index into the prior run): largest pi, lowest index among equals; 0xFFFF = none left
    double norm;    // np.sum(Ps[s]) that renormalises the masked prior (:90-92), or the fallback's sum (:100)
};
#define HF_EXPANDED 1u
#define HF_FALLBACK 2u
__host__ __device__ inline u32 hdr_term_kind(const NodeHdr &h) { return (h.flags >> 2) & 3u; }

// Instance pool of auto_restart, in DEVICE memory: kernel arguments are baked into captured HIP graphs, the pool changes between
// pools of one BatchedSelfPlay (size, episode-id base, even the buffers when it grows), so restarting slots read it from here.
struct PoolDesc {
    long long n_instances;
    unsigned long long first_id;
    const unsigned char *wh;  // [n_instances][N][2]
    const int *area, *max_h;
};
// A visited edge: one 32-byte record = two 16-byte loads.  (Five parallel arrays -- idx, N, child, Q, P -- cost a node with a few visited
// edges five cache lines per selection and a backup two; the record costs one.)
struct VisEntry {
    double q;    // Qsa
    double p;    // Ps[s][a] = float64(pi) / norm
    u32 n;       // Nsa (low 30 bits) | kind of Q (top 2 bits)
    u32 child;   // child node id (NONE32 until first traversed)
    u32 idx;     // index into the node's legal-move run
    u32 pad;
};
struct DP {  // device view of a context, passed by value to every kernel
    int W, H, N, A, G, sims, node_cap, edge_cap, vis_cap, table_cap, KW, RW, RMW, move_rule;
    int reclaim;   // 1: a level's legal-move runs and visited blocks are recycled once the root has moved past it
    int pchunk, n_pchunks, vchunk, n_vchunks;  // level arenas: entries per chunk, chunks per slot
    int step_cap;  // max simulations a slot runs in one k_search launch (0 = until it needs the evaluator)
    u32 magicW;  // a / W == (a * magicW) >> 20 for a < 8192
    double cpuct;
    u64 seed, tie_salt;
    // per slot
    u8 *item_wh;  // [G][N][2]
    int *total_area, *max_h;
    double *bl;     // R2 threshold snapshot taken when the episode began
    int *has_buf;
    u32 *root, *n_nodes;
    // level arenas (one for legal-move runs, one for visited blocks): per slot and level the chunk being filled, its fill, the
    // level's chunk list; per slot a stack of recycled chunks and the count of never-used ones
    u16 *pa_cur, *pa_head, *pa_next, *pa_stack, *va_cur, *va_head, *va_next, *va_stack;
    u32 *pa_used, *pa_tf, *va_used, *va_tf;
    u32 *peak_chunks;  // [G][2] per-slot high-water marks (sizing aid)
    int *phase, *sims_done, *moves;
    u64 *episode;
    u32 *leaf_node;
    int *path_len;
    u32 *path_edge, *path_node;  // [G][N]  (visited-entry index, node) per level
    int *game_row;
    int *last_outcome;
    double *last_score;
    double *last_v;     // value returned by the slot's latest simulation (MCTS.search's return value)
    int *last_vkind;
    // evaluation rows: identity (row b = slot b, used by captured graphs: no atomics, fixed shapes) or the compact list that
    // k_compact builds in slot order when the host asks for the number of waiting leaves
    int rows_identity;
    int *eval_count, *eval_slot;
    // arenas: ONE slab per slot (slab_stride bytes apart) holding the slot's ten regions back to back, hottest first -- a wave's
    // accesses then fall into 2-3 translation fragments instead of ten (UTCL2 was busy 76 % of k_search with one array per
    // region across all slots); the pointers below are slot 0's regions, slot g's are slab_stride * g bytes further
    size_t slab_stride;
    NodeHdr *hdr;   // [G][node_cap]
    u32 *key;       // [G][node_cap][KW]
    u16 *pAct;      // [G][edge_cap]  legal moves of every node, ascending action
    float *pPi;     // [G][edge_cap]  evaluator probability of that move (NNet.predict's pi, float32)
    VisEntry *vis;  // [G][vis_cap]   visited edges
    u64 *table;     // [G][table_cap]  (tag << 32) | (node id + 1), 0 = empty
    // NumPy pairwise-sum plan over A elements, carried in the kernel arguments (scalar loads)
    int n_leaves;
    u16 leaf_lo[MAX_LEAVES];
    u8 leaf_n[MAX_LEAVES], sched_dst[MAX_LEAVES], sched_src[MAX_LEAVES];
    // global
    double *g_bl;
    int *g_has_buf;
    u64 *counters;   // [CNT_N] totals handed to the host (k_reduce_counters)
    u64 *slot_cnt;   // [G][CNT_N] per-slot event counts: plain updates by the slot's own wave, no hot atomics
    int *error;
    // instance pool for auto_restart
    int auto_restart;
    int onehot_examples;  // greedy self-play records pi as a one-hot on the chosen action (MCTS_bpp.py:43-49)
    long long n_instances;
    u64 first_id;
    const u8 *pool_wh;      // [n_instances][N][2]
    const int *pool_area, *pool_max_h;
    const PoolDesc *pool_desc;  // device copy of the five fields above (what the kernels read)
    unsigned long long *next_instance;
    // replay buffer (CoachBPP.executeEpisode's trainExamples, CoachBPP.py:80,99)
    long long max_examples;
    unsigned long long *ex_count;
    u32 *ex_key;      // [max_examples][KW]
    u8 *ex_wh;        // [max_examples][N][2]
    // root visit counts, SPARSE: one (action, count) pair per VISITED root edge (2 % of the action space at 20x20/32: ~0.2 KB per
    // example instead of 4 A = 2.5 KB), appended to one pool; an example holds its first entry and its entry count
    u32 *ex_sp_off, *ex_sp_n;  // [max_examples]
    u16 *ex_sp_act;            // [sp_cap]
    u32 *ex_sp_cnt;            // [sp_cap]
    unsigned long long *ex_sp_cursor;
    long long sp_cap;
    int *ex_value;    // ranked outcome of the episode, 0 until it ends
    u64 *ex_episode;  // episode id and move number of every example: the buffer fills in completion order across slots,
    int *ex_move;     // the reference appends episode by episode, move by move (CoachBPP.py:80,133)
    u32 *slot_ex;     // [G][N] example indices of the running episode
    // evaluator stem (first convolution + max-pool computed from the packed state)
    int *stemT, *stemTB, *stemBias;  // [N][25][16], [512][16], [16]: tap sums in per-channel fixed point (k_stem_tables)
    float *stemScale;                // [16] value of one fixed-point unit of channel o (a power of two)
    int Hp, Wp;
    // finished-episode ring
    int fin_cap;
    int *fin_count;
    u64 *fin_episode;
    int *fin_outcome, *fin_moves;
    double *fin_score;
};
// the first error of a launch sequence is the one reported (later ones are usually its consequences)
__device__ __forceinline__ void set_error(const DP &p, int code) { atomicCAS(p.error, 0, code); }

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// index of this wave in its workgroup, as a wave-uniform (SGPR) value: everything derived from it -- the slot id, the slot's region
// pointers -- then lives in scalar registers; derived from threadIdx.x alone the compiler keeps all of it per lane
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// A value every lane of the wave holds (loaded from a wave-uniform address, broadcast by lane 0 ...) moved to scalar registers: the
// compiler cannot prove that a vector load returned one value, so without this every branch, loop bound and address derived
// from a node header or a slot's state is per-lane VALU work behind exec masks.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uni_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ u64 uni(u64 v) { return ((u64)uni((u32)(v >> 32)) << 32) | (u64)uni((u32)v); }
__device__ __forceinline__ void wave_sync() {
    // LDS / global accesses of one wave are issued in order; this only stops the compiler from moving
    // memory operations across the point and makes earlier stores visible to the other lanes.
    // (A wavefront-scope fence -- no s_waitcnt vmcnt(0) -- passed every parity test and measured the same when it was tried; the
    // workgroup scope stays because the tree kernels re-read, by other lanes, global data a lane has just stored.)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}
// The same for data a wave exchanges with itself through LDS only (the stage kernels' images and staging rows): LDS operations of one
// wave execute in order, so nothing has to be waited for -- the wavefront-scope fence only keeps the compiler from moving LDS accesses
// across the point.  wave_sync()'s workgroup fence is an s_waitcnt vmcnt(0) as well: in a persistent MFMA kernel that parks the wave
// until its prefetched next input has arrived from HBM and its output stores have drained, four times per task.
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Workgroup barrier for LDS traffic only: every wave's LDS operations have completed (lgkmcnt), then s_barrier.  __syncthreads() also
// waits for the wave's global loads and stores (vmcnt(0)), which a persistent kernel wants to keep in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ u64 lanes_below() { return (1ull << lane_id()) - 1ull; }
__device__ __forceinline__ u64 wave_xor_u64(u64 v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v ^= __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// Wave-wide scans on the DPP path: one VALU instruction per step (row_shr:1/2/4/8 inside the 16-lane rows, row_bcast:15 and
// row_bcast:31 across them) instead of a ds_bpermute + select + op per step.  Lanes a step has no source for read 0, the identity of
// OR and of unsigned ADD, so no per-step predicate is needed.  All 64 lanes must be active.
template <int CTRL, int ROWMASK> __device__ __forceinline__ u32 dpp_or_zero(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ u32 wave_scan_or(u32 v) {  // inclusive, ascending lanes
    v |= dpp_or_zero<0x111, 0xf>(v); v |= dpp_or_zero<0x112, 0xf>(v); v |= dpp_or_zero<0x114, 0xf>(v); v |= dpp_or_zero<0x118, 0xf>(v);
    v |= dpp_or_zero<0x142, 0xa>(v); v |= dpp_or_zero<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ u64 wave_scan_or(u64 v) {
    return ((u64)wave_scan_or((u32)(v >> 32)) << 32) | wave_scan_or((u32)v);
}
__device__ __forceinline__ u32 wave_scan_add(u32 v) {  // inclusive, ascending lanes
    v += dpp_or_zero<0x111, 0xf>(v); v += dpp_or_zero<0x112, 0xf>(v); v += dpp_or_zero<0x114, 0xf>(v); v += dpp_or_zero<0x118, 0xf>(v);
    v += dpp_or_zero<0x142, 0xa>(v); v += dpp_or_zero<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ u32 wave_shift_up1(u32 v) { return dpp_or_zero<0x138, 0xf>(v); }  // lane l reads lane l - 1 (wave_shr:1), lane 0 reads 0
__device__ __forceinline__ u64 wave_shift_up1(u64 v) { return ((u64)wave_shift_up1((u32)(v >> 32)) << 32) | wave_shift_up1((u32)v); }
__device__ __forceinline__ u64 mix64(u64 x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline u64 full_mask(int w) { return w >= 64 ? ~0ull : ((1ull << w) - 1ull); }

template <typename row_t> struct RowOps;
template <> struct RowOps<u32> {
    static __device__ __forceinline__ u32 at(u32 mine, int r) { return (u32)__builtin_amdgcn_readlane((int)mine, r); }
    static __device__ __forceinline__ int popc(u32 x) { return __popc(x); }
};
template <> struct RowOps<u64> {
    static __device__ __forceinline__ u64 at(u64 mine, int r) {
        u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)mine, r);
        u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(mine >> 32), r);
        return ((u64)hi << 32) | lo;
    }
    static __device__ __forceinline__ int popc(u64 x) { return __popcll(x); }
};

// bits j .. j+w-1, clipped at W exactly like the NumPy slice board[r, j:j+w]
template <typename row_t> __device__ __forceinline__ row_t window(int j, int w, int W) {
    return (row_t)((full_mask(w) << j) & full_mask(W));
}

// ------------------------------------------------------------------------------------------------
// game rules on the register-resident board
// ------------------------------------------------------------------------------------------------
struct ValidSink {  // where gen_valid_moves puts its result
    u16 *act;       // compact legal-move list (or null)
    u8 *mask;       // dense 0/1 mask of A bytes (or null)
    int cap;        // list entries available
    u64 *vm;        // LDS scratch of the wave: [128] per-item column masks
    int have_sizes, w_lo, h_lo, w_hi, h_hi;  // item sizes of lanes' items (i, 64 + i) when the caller loaded them already
};
#define VM_WORDS 128

// BinPackingGame.getValidMoves (BinPackingGame.py:78-92 -> BinPackingLogic.py:80-93 and :47-78) for one state per wave.
// Whether (item i, column j) is legal depends on the item's SIZE only, and the two conditions separate:
//   adjacency  (:63-70)  depends on the width w:   j == 0 or cell (t, j - 1) occupied, t = first row whose window [j, j + w) is
//                         empty (H - 1 if none).  Lane r holds E_w(r) = columns whose window is empty in row r (E_1 = ~row,
//                         E_w = E_(w-1) & (E_1 >> (w - 1)): two instructions per width for all rows at once); an exclusive prefix-OR
//                         over the lanes leaves in each lane the columns whose FIRST empty row it is, one AND with (row << 1) tests
//                         the left neighbour, an OR-reduction gives the width's adjacency mask -- ~30 instructions per width that
//                         an unplaced item actually has, whatever W and H;
//   area       (:89)     depends on (w, h):   cells occupied in columns j .. j + w - 1  <=  w (H - h).  Lane c counts column c
//                         once per state, an inclusive prefix sum turns a window's count into one subtraction per width, and an
//                         item's mask is one compare + ballot against its threshold.
// The per-item masks (W bits) go to LDS; then lanes = items: popcounts, a prefix sum for the offsets, and every lane writes its
// item's actions i * W + j in ascending order.  The previous form tested every (item, column) pair against all H rows:
// ~5 H instructions per 64 actions, 1 100 per node at 20x20/32 and 25 000 at 50x50/128.
// Returns the number of legal moves, or -1 if they do not fit sink.cap.  Must be called by all 64 lanes.
// BIG = false: the instance for N <= 64 -- the second item word and everything derived from it folds away.
template <typename row_t, bool BIG = true>
__device__ int gen_valid_moves(const DP &p, const u8 *wh, row_t myrow, u64 rem0, u64 rem1, const ValidSink &sink) {
    const int lane = lane_id(), W = p.W, H = p.H, N = p.N;
    const bool big = BIG && N > 64;
    const row_t full = (row_t)full_mask(W);
    int w_lo = sink.w_lo, h_lo = sink.h_lo, w_hi = sink.w_hi, h_hi = sink.h_hi;
    const bool un_lo = lane < N && ((rem0 >> lane) & 1ull), un_hi = big && lane + 64 < N && ((rem1 >> lane) & 1ull);  // plane sum != 0 (BinPackingGame.py:86)
    if (!sink.have_sizes) {
        w_lo = h_lo = w_hi = h_hi = 0;
        if (lane < N) { w_lo = wh[2 * lane]; h_lo = wh[2 * lane + 1]; }
        if (big && lane + 64 < N) { w_hi = wh[2 * (lane + 64)]; h_hi = wh[2 * (lane + 64) + 1]; }
    }
    sink.vm[lane] = 0ull;
    if (big) sink.vm[64 + lane] = 0ull;
    // occupied cells per column (lane c <-> column c), inclusive prefix sum I and its left neighbour
    int I = 0;
    for (int r = 0; r < H; ++r) {
        const row_t rr = RowOps<row_t>::at(myrow, r);
        I += lane < W ? (int)((rr >> lane) & 1) : 0;
    }
    I = (int)wave_scan_add((u32)I);
    const int Im1 = (int)wave_shift_up1((u32)I);
    const row_t E1 = lane < H ? (row_t)(~myrow & full) : (row_t)0, left = (row_t)(myrow << 1);  // bit j of `left`: cell (r, j - 1) occupied
    row_t E = E1;
    for (int w = 1; w <= W; ++w) {
        if (w > 1) E = E & (row_t)(E1 >> (w - 1));  // bit j: the window [j, j + w) of this lane's row is empty (and fits: E1 has no bits >= W)
        const u64 im0 = __ballot(un_lo && w_lo == w), im1 = big ? __ballot(un_hi && w_hi == w) : 0ull;
        if ((im0 | im1) == 0ull) continue;  // no unplaced item of this width
        const row_t X = wave_scan_or(E);                      // inclusive prefix-OR over the rows
        const row_t Xex = wave_shift_up1(X);
        const row_t anyrow = RowOps<row_t>::at(X, 63);        // columns with an empty window in SOME row
        row_t G = (row_t)(E & ~Xex) & left;                   // first empty row of the column is this one, and its left neighbour is occupied
        if (lane == H - 1) G |= (row_t)(~anyrow & full) & left;  // no empty row: the for / break falls through with t = H - 1 (:66-68)
        G = RowOps<row_t>::at(wave_scan_or(G), 63);           // OR over all rows
        const u64 adj = (u64)((G | (row_t)1) & (row_t)full_mask(W - w + 1));  // j == 0 is always adjacent; for j in range(W - w + 1) (:87)
        const int S = __shfl(I, (lane + w - 1) & 63) - Im1;  // cells occupied in columns lane .. lane + w - 1
        for (u64 m = im0; m; m &= m - 1) {
            const int i = __ffsll((long long)m) - 1, h = __builtin_amdgcn_readlane(h_lo, i);
            const u64 v = __ballot(S <= w * H - w * h) & adj;
            if (lane == 0) sink.vm[i] = v;
        }
        for (u64 m = im1; m; m &= m - 1) {
            const int i = __ffsll((long long)m) - 1, h = __builtin_amdgcn_readlane(h_hi, i);
            const u64 v = __ballot(S <= w * H - w * h) & adj;
            if (lane == 0) sink.vm[64 + i] = v;
        }
    }
    wave_sync();
    // lanes = items: counts, offsets, actions in ascending order
    u64 m_lo = sink.vm[lane], m_hi = big ? sink.vm[64 + lane] : 0ull;
    int inc_lo = __popcll(m_lo), inc_hi = __popcll(m_hi);
    const int c_lo = inc_lo, c_hi = inc_hi;
    inc_lo = (int)wave_scan_add((u32)inc_lo);
    if (big) inc_hi = (int)wave_scan_add((u32)inc_hi);
    const int tot_lo = __builtin_amdgcn_readlane(inc_lo, 63), nv = tot_lo + __builtin_amdgcn_readlane(inc_hi, 63);
    if (sink.mask)
        for (int a = lane; a < p.A; a += 64) sink.mask[a] = 0;
    if (sink.act && nv > sink.cap) return -1;
    wave_sync();
    int pos = inc_lo - c_lo;
    for (; m_lo; m_lo &= m_lo - 1) {
        const int a = lane * W + (__ffsll((long long)m_lo) - 1);
        if (sink.act) sink.act[pos] = (u16)a;
        if (sink.mask) sink.mask[a] = 1;
        ++pos;
    }
    pos = tot_lo + inc_hi - c_hi;
    for (; m_hi; m_hi &= m_hi - 1) {
        const int a = (64 + lane) * W + (__ffsll((long long)m_hi) - 1);
        if (sink.act) sink.act[pos] = (u16)a;
        if (sink.mask) sink.mask[a] = 1;
        ++pos;
    }
    wave_sync();
    return nv;
}

// Bin.execute_move (BinPackingLogic.py:95-109): rows are scanned bottom-up (row 0 first), every row
// whose window is empty gets the window filled until h rows are filled.  Lane r owns row r: the
// ballot of "window empty" marks the candidate rows, the first h of them (prefix popcount) fill.
template <typename row_t>
__device__ __forceinline__ row_t apply_move_rows(row_t myrow, int H, int W, int j, int w, int h) {
    const row_t M = window<row_t>(j, w, W);
    bool empty = (lane_id() < H) && ((myrow & M) == 0);
    u64 m = __ballot(empty);
    if (empty && __popcll(m & lanes_below()) < h) myrow |= M;
    return myrow;
}

// BinPackingGame.getRankedReward (BinPackingGame.py:188-212) + get_minimal_bin_height (:181-186).
// Returns +1 / -1 or 2 for the r == bl tie; *r_out = r.
template <typename row_t>
__device__ int ranked_reward(row_t myrow, int H, int W, int total_area, int max_h, bool has_buf, double bl, double *r_out) {
    int cells = wave_sum_i32(lane_id() < H ? RowOps<row_t>::popc(myrow) : 0);
    u64 occ = __ballot(lane_id() < H && myrow != 0);
    int top = occ ? 64 - __clzll(occ) : 1;  // 1 + highest occupied row; 1 on an empty grid
    double r;
    if (cells != total_area) {
        r = 0.0;  // :193-195
    } else {
        double need = ceil((double)total_area / (double)W);  // np.ceil(area / W)
        double best = need >= (double)max_h ? need : (double)max_h;
        r = best / (double)top;  // :198
    }
    *r_out = r;
    if (!has_buf) return 1;            // :203-204
    if (r > bl || r == 1.0) return 1;  // :207-208
    if (r < bl) return -1;             // :209-210
    return 2;                          // :211-212
}

// deterministic stand-in for np.random.choice([1,-1]) (BinPackingGame.py:212): sequential splitmix64 over
// the row masks and the remaining-item words (tests/evaluators.py tie_value computes the same).
template <typename row_t>
__device__ int tie_value(row_t myrow, int H, int N, u64 rem0, u64 rem1, u64 salt) {
    u64 h = mix64(salt);
    for (int r = 0; r < H; ++r) h = mix64(h ^ (u64)RowOps<row_t>::at(myrow, r));
    h = mix64(h ^ rem0);
    if (N > 64) h = mix64(h ^ rem1);
    return (mix64(h ^ 0x7469ull) & 1ull) ? 1 : -1;
}

// ------------------------------------------------------------------------------------------------
// Q update: (Nsa*Qsa + v)/(Nsa+1) of MCTS_bpp.py:131 under NumPy 2 promotion (see oracle/rp_oracle.c
// orc_q_update for the derivation).  Q is stored as a double plus a 2-bit kind.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void q_update(double &q, u32 &kind, u32 n, double v, u32 vkind) {
    if (n == 0) { q = v; kind = vkind; return; }  // :135
    const double dn = (double)n, dn1 = (double)(n + 1);
    if (kind == RP_KIND_WEAK) {
        double nq = dn * q;
        if (vkind == RP_KIND_F32) {
            float s = (float)nq + (float)v;
            q = (double)(s / (float)dn1);
            kind = RP_KIND_F32;
        } else {
            q = (nq + v) / dn1;
            kind = vkind;
        }
    } else if (kind == RP_KIND_F32) {
        float nq = (float)dn * (float)q;
        if (vkind == RP_KIND_F64) {
            q = ((double)nq + v) / dn1;
            kind = RP_KIND_F64;
        } else {
            float s = nq + (float)v;
            q = (double)(s / (float)dn1);
        }
    } else {
        q = (dn * q + v) / dn1;
    }
}

// ------------------------------------------------------------------------------------------------
// Level arena: bump allocation in fixed-size chunks, one open chunk per game level (number of items placed).  A state of
// level d can only be reached while the root's level is below d, so once the root has moved past a level every run / block
// of that level is dead and its chunks go back to the slot's stack (DP::reclaim).  All bookkeeping is done by lane 0 of the
// slot's wave; the result is broadcast.
// ------------------------------------------------------------------------------------------------
struct Arena {
    u32 chunk, n_chunks;
    u16 *cur, *head, *next, *stack;  // cur/head: [N+1] per level; next/stack: [n_chunks]
    u32 *used;                       // [N+1]
    u32 *tf;                         // [0] chunks on the stack, [1] chunks never used yet
};
// offset of a run that can hold `need` entries at level d (the level's open chunk, or a new one); NONE32 when the arena is full
__device__ u32 arena_reserve(const Arena &a, int d, u32 need, u32 *room) {
    u32 off = NONE32, left = 0;
    if (lane_id() == 0) {
        u32 c = a.cur[d], used = a.used[d];
        if (c == 0xFFFFu || used + need > a.chunk) {
            u32 top = a.tf[0];
            if (top > 0) { c = a.stack[top - 1]; a.tf[0] = top - 1; }
            else { u32 f = a.tf[1]; if (f < a.n_chunks) { c = f; a.tf[1] = f + 1; } else c = 0xFFFFu; }
            if (c != 0xFFFFu) { a.next[c] = a.head[d]; a.head[d] = (u16)c; a.cur[d] = (u16)c; a.used[d] = 0; used = 0; }
        }
        if (c != 0xFFFFu && need <= a.chunk) { off = c * a.chunk + used; left = a.chunk - used; }
    }
    *room = uni(left);  // lane 0 did the bookkeeping
    return uni(off);
}
__device__ void arena_commit(const Arena &a, int d, u32 n) {
    if (lane_id() == 0) a.used[d] += n;
}
__device__ void arena_free_level(const Arena &a, int d) {
    if (lane_id() == 0) {
        u32 c = a.head[d], top = a.tf[0];
        while (c != 0xFFFFu) { a.stack[top++] = (u16)c; c = a.next[c]; }
        a.tf[0] = top;
        a.head[d] = 0xFFFFu; a.cur[d] = 0xFFFFu; a.used[d] = 0;
    }
}
// A never-used chunk is taken only while the stack of recycled ones is empty, so the high-water mark of chunks in use is tf[1]
// itself: it is folded into `peak` when an episode's arenas are reset (and read live by k_reduce_peaks), not after every launch.
__device__ void arena_reset(const Arena &a, int levels, u32 *peak) {
    for (int d = lane_id(); d < levels; d += 64) { a.cur[d] = 0xFFFFu; a.head[d] = 0xFFFFu; a.used[d] = 0; }
    if (lane_id() == 0) {
        const u32 hw = a.tf[1];
        if (hw > *peak) *peak = hw;
        a.tf[0] = 0; a.tf[1] = 0;
    }
}

// region pointer of slot g (DP::slab_stride)
template <typename T> __device__ __host__ __forceinline__ T *slot_region(const DP &p, T *slot0, int g) {
    return (T *)((u8 *)slot0 + (size_t)g * p.slab_stride);
}

// ------------------------------------------------------------------------------------------------
// slot-local tree
// ------------------------------------------------------------------------------------------------
template <typename row_t, bool BIG = true> struct Tree {
    const DP &p;
    int g;
    NodeHdr *hdr;
    u32 *key;
    u16 *pAct;
    float *pPi;
    VisEntry *vis;
    u64 *table;
    const u8 *wh;
    Arena pa, va;  // legal-move runs, visited blocks
    u32 n_nodes;
    u32 cnt = 0;   // per-launch event counts: LANE k holds counter k (one VGPR; sixteen wave-uniform counters took sixteen SGPRs of a
                   // kernel that already spills scalar registers) -- count() adds, add_counters() flushes
    u16 *stage;    // this wave's LDS staging run of A actions (kernels that can create nodes), else null
    u64 *vm;       // this wave's LDS scratch for gen_valid_moves ([VM_WORDS]), with `stage`
    u32 *vmask;    // this wave's LDS bit mask over a node's legal moves ([MAX_MASK_WORDS]; kernels that can add visited edges), else null
    int have_sizes = 0, w_lo = 0, h_lo = 0, w_hi = 0, h_hi = 0;  // the slot's item sizes in lanes (load_sizes), constant over an episode

    __device__ Tree(const DP &p_, int g_, u16 *stage_ = nullptr, u64 *vm_ = nullptr, u32 *vmask_ = nullptr) : p(p_), g(g_), stage(stage_), vm(vm_), vmask(vmask_) {
        hdr = slot_region(p, p.hdr, g);
        key = slot_region(p, p.key, g);
        pAct = slot_region(p, p.pAct, g); pPi = slot_region(p, p.pPi, g);
        vis = slot_region(p, p.vis, g);
        table = slot_region(p, p.table, g);
        wh = p.item_wh + (size_t)g * p.N * 2;
        n_nodes = p.n_nodes[g];
        const size_t lv = (size_t)g * (p.N + 1);
        pa.chunk = p.pchunk; pa.n_chunks = p.n_pchunks;
        pa.cur = p.pa_cur + lv; pa.head = p.pa_head + lv; pa.used = p.pa_used + lv;
        pa.next = p.pa_next + (size_t)g * p.n_pchunks; pa.stack = p.pa_stack + (size_t)g * p.n_pchunks; pa.tf = p.pa_tf + (size_t)g * 2;
        va.chunk = p.vchunk; va.n_chunks = p.n_vchunks;
        va.cur = p.va_cur + lv; va.head = p.va_head + lv; va.used = p.va_used + lv;
        va.next = p.va_next + (size_t)g * p.n_vchunks; va.stack = p.va_stack + (size_t)g * p.n_vchunks; va.tf = p.va_tf + (size_t)g * 2;
    }
    __device__ __forceinline__ void count(int k, u32 v = 1u) { cnt += lane_id() == k ? v : 0u; }
    // one coalesced read-modify-write of the slot's sixteen 64-bit totals (plain stores: the slot's own wave is their only writer)
    __device__ void flush_counters() {
        const int lane = lane_id();
        if (lane < CNT_N && cnt) p.slot_cnt[(size_t)g * CNT_N + lane] += (u64)cnt;
        cnt = 0;
    }
    __device__ void store_sizes() {
        if (lane_id() == 0) p.n_nodes[g] = n_nodes;
    }
    // requested at kernel start, next to the slot's other state, instead of as a dependent read inside every node creation
    __device__ void load_sizes() {
        const int lane = lane_id();
        if (lane < p.N) { w_lo = wh[2 * lane]; h_lo = wh[2 * lane + 1]; }
        if (BIG && lane + 64 < p.N) { w_hi = wh[2 * (lane + 64)]; h_hi = wh[2 * (lane + 64) + 1]; }
        have_sizes = 1;
    }
    __device__ void reset_arenas() {
        arena_reset(pa, p.N + 1, p.peak_chunks + (size_t)g * 2);
        arena_reset(va, p.N + 1, p.peak_chunks + (size_t)g * 2 + 1);
    }
    // a node's header as wave-uniform scalars (one 32-byte vector load, eight readfirstlanes)
    __device__ __forceinline__ NodeHdr load_hdr(u32 node) const {
        static_assert(sizeof(NodeHdr) == 32, "NodeHdr is eight dwords");
        const uint4 *q = (const uint4 *)(hdr + node);
        const uint4 a = q[0], b = q[1];
        u32 w[8] = {uni(a.x), uni(a.y), uni(a.z), uni(a.w), uni(b.x), uni(b.y), uni(b.z), uni(b.w)};
        NodeHdr h;
        __builtin_memcpy(&h, w, sizeof h);
        return h;
    }
    static __device__ __forceinline__ int level_of(int N, u64 rem0, u64 rem1) { return N - __popcll(rem0) - __popcll(rem1); }

    // key of a node -> lane-resident rows + uniform remaining words.  The key is H rows followed by the remaining-item words, so ONE
    // load -- lane l takes key element l -- brings all of it when H + words <= 64 (packed() below); fetch_key only requests it, so a
    // caller can ask for a node's key together with its header and decode it a round trip later, if at all (resolve_child).
    static constexpr int KEY_PER_ROW = sizeof(row_t) / 4;
    __device__ __forceinline__ bool packed() const { return p.H + (p.RMW + KEY_PER_ROW - 1) / KEY_PER_ROW <= 64; }
    __device__ __forceinline__ row_t fetch_key(u32 node) const {
        const row_t *k = (const row_t *)(key + (size_t)node * p.KW);
        const int n = p.H + (p.RMW + KEY_PER_ROW - 1) / KEY_PER_ROW;
        return lane_id() < n ? k[lane_id()] : (row_t)0;
    }
    __device__ __forceinline__ void decode_key(row_t kv, row_t &myrow, u64 &rem0, u64 &rem1) const {
        myrow = lane_id() < p.H ? kv : (row_t)0;
        if (KEY_PER_ROW == 2) {
            rem0 = (u64)RowOps<row_t>::at(kv, p.H);
            if (p.RMW < 2) rem0 &= 0xFFFFFFFFull;  // an odd word count leaves a padding word behind the last one
            rem1 = BIG && p.RMW > 2 ? (u64)RowOps<row_t>::at(kv, p.H + 1) : 0ull;
            if (BIG && p.RMW == 3) rem1 &= 0xFFFFFFFFull;
        } else {
            rem0 = (u64)(u32)RowOps<row_t>::at(kv, p.H);
            if (p.RMW > 1) rem0 |= (u64)(u32)RowOps<row_t>::at(kv, p.H + 1) << 32;
            rem1 = 0;
            if (BIG && p.RMW > 2) rem1 = (u64)(u32)RowOps<row_t>::at(kv, p.H + 2);
            if (BIG && p.RMW > 3) rem1 |= (u64)(u32)RowOps<row_t>::at(kv, p.H + 3) << 32;
        }
    }
    __device__ void load_key(u32 node, row_t &myrow, u64 &rem0, u64 &rem1) const {
        if (packed()) { decode_key(fetch_key(node), myrow, rem0, rem1); return; }
        const u32 *k = key + (size_t)node * p.KW;
        myrow = 0;
        if (lane_id() < p.H) myrow = ((const row_t *)k)[lane_id()];
        const u32 *rw = k + p.H * p.RW;
        rem0 = rw[0];
        if (p.RMW > 1) rem0 |= (u64)rw[1] << 32;
        rem1 = 0;
        if (p.RMW > 2) rem1 = rw[2];
        if (p.RMW > 3) rem1 |= (u64)rw[3] << 32;
        rem0 = uni(rem0); rem1 = uni(rem1);
    }
    __device__ void store_key(u32 node, row_t myrow, u64 rem0, u64 rem1) {
        u32 *k = key + (size_t)node * p.KW;
        if (lane_id() < p.H) ((row_t *)k)[lane_id()] = myrow;
        u32 *rw = k + p.H * p.RW;
        if (lane_id() < p.RMW) rw[lane_id()] = (u32)((lane_id() < 2 ? rem0 : rem1) >> (32 * (lane_id() & 1)));
    }
    __device__ bool key_equals(u32 node, row_t myrow, u64 rem0, u64 rem1) {
        row_t r2; u64 a0, a1;
        load_key(node, r2, a0, a1);
        count(CNT_KEYBYTES, (u32)p.KW * 4u);
        return __all((r2 == myrow) && (a0 == rem0) && (a1 == rem1)) != 0;
    }
    __device__ u64 hash_state(row_t myrow, u64 rem0, u64 rem1) const {
        u64 h = lane_id() < p.H ? mix64((u64)myrow + (u64)(lane_id() + 1) * 0x9E3779B97F4A7C15ull) : 0ull;
        h = wave_xor_u64(h);
        h ^= mix64(rem0 ^ 0xD1B54A32D192ED03ull) ^ mix64(rem1 + 0x8CB92BA72F3D8DD7ull);
        return mix64(h);
    }
    // s in self.Es ?  Open addressing over ALIGNED buckets of 16 slots = one 128-byte line per probe (lanes 16-63 repeat the addresses of
    // lanes 0-15: the same request); at the table's load of <= 1/2 a bucket without an empty slot is a ~1e-3 event, so a lookup is
    // one line where the 64-slot window took four.  A key lives in the first bucket from its home that had room when it was inserted;
    // nothing is ever deleted, so the first empty slot ends the search.  Returns node id or NONE32 (+ the slot to insert at).
    __device__ u32 find(row_t myrow, u64 rem0, u64 rem1, u64 h, u32 &insert_slot) {
        const u32 mask = (u32)p.table_cap - 1u, tag = (u32)(h >> 32);
        u32 slot = (u32)h & mask & ~15u;
        for (int it = 0; it <= p.table_cap / 16; ++it) {
            u64 e = table[slot + (lane_id() & 15)];
            count(CNT_PROBES);
            u64 empty = __ballot(e == 0ull) & 0xFFFFull;
            u64 hit = __ballot(e != 0ull && (u32)(e >> 32) == tag) & 0xFFFFull;
            int first_empty = empty ? __ffsll((long long)empty) - 1 : 16;
            u64 cand = hit & ((1ull << first_empty) - 1ull);
            while (cand) {
                int l = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                u32 id = (u32)__builtin_amdgcn_readlane((int)(u32)e, l) - 1u;
                if (key_equals(id, myrow, rem0, rem1)) return id;
            }
            if (first_empty < 16) { insert_slot = slot + first_empty; return NONE32; }
            slot = (slot + 16) & mask;
        }
        if (lane_id() == 0) set_error(p, ERR_TABLE_FULL);
        insert_slot = NONE32;
        return NONE32;
    }
    // A state seen for the first time: the Es / Vs part of MCTS.search (MCTS_bpp.py:78-79 getGameEnded ->
    // has_valid_moves / getRankedReward; :88 getValidMoves).  Allocates the node, stores its key, its legal
    // moves as a run of actions (their priors are filled in by the commit kernel after the evaluator ran) or its
    // terminal value.  Returns the node id or NONE32 on arena overflow.
    __device__ u32 materialize(row_t myrow, u64 rem0, u64 rem1, u64 h, u32 insert_slot) {
        if (n_nodes >= (u32)p.node_cap || insert_slot == NONE32) {
            if (lane_id() == 0) set_error(p, ERR_NODE_CAP);
            return NONE32;
        }
        const u32 id = n_nodes;
        store_key(id, myrow, rem0, rem1);
        const int level = level_of(p.N, rem0, rem1);
        // The legal moves go to the wave's LDS staging run first, so the arena is asked for exactly as many entries as the node
        // has (a run reserved for the worst case -- one move per unplaced item and column -- opened a fresh chunk per node on
        // large boards: 460 of 6 400 entries used at 50x50/128).
        ValidSink sink;
        sink.act = stage; sink.mask = nullptr; sink.cap = p.A; sink.vm = vm;
        sink.have_sizes = have_sizes; sink.w_lo = w_lo; sink.h_lo = h_lo; sink.w_hi = w_hi; sink.h_hi = h_hi;
        const int nv = gen_valid_moves<row_t, BIG>(p, wh, myrow, rem0, rem1, sink);
        wave_sync();
        u32 room, off = 0;
        if (nv > 0) {
            off = arena_reserve(pa, level, (u32)nv, &room);
            if (off == NONE32) {
                if (lane_id() == 0) set_error(p, ERR_EDGE_CAP);
                return NONE32;
            }
            for (int k = lane_id(); k < nv; k += 64) pAct[off + k] = stage[k];
            arena_commit(pa, level, (u32)nv);
        }
        NodeHdr hd;
        hd.ns = 0; hd.prior_off = off; hd.vis_off = 0; hd.n_valid = (u16)nv; hd.vis_n = 0; hd.vis_cap = 0; hd.term = 0;
        hd.flags = (u8)(RP_KIND_WEAK << 2); hd.depth = (u8)level; hd.pad = 0; hd.best_k = 0xFFFFu; hd.norm = 0.0;
        if (nv == 0) {  // no legal move: game over (BinPackingGame.py:112-114)
            double r;
            int e = ranked_reward<row_t>(myrow, p.H, p.W, p.total_area[g], p.max_h[g], p.has_buf[g] != 0, p.bl[g], &r);
            if (e == 2) { e = tie_value<row_t>(myrow, p.H, p.N, rem0, rem1, p.tie_salt); hd.flags = (u8)(RP_KIND_F64 << 2); }
            hd.term = (int8_t)e;
        }
        if (lane_id() == 0) {
            hdr[id] = hd;
            table[insert_slot] = ((u64)(u32)(h >> 32) << 32) | (u64)(id + 1u);
        }
        n_nodes++;
        count(CNT_NODES);
        return id;
    }
    __device__ u32 find_or_materialize(row_t myrow, u64 rem0, u64 rem1, bool *was_new) {
        u64 h = hash_state(myrow, rem0, rem1);
        u32 slot;
        u32 id = find(myrow, rem0, rem1, h, slot);
        *was_new = false;
        if (id == NONE32) { id = materialize(myrow, rem0, rem1, h, slot); *was_new = true; }
        return id;
    }
    // child state of (node, action): BinPackingGame.getNextState (BinPackingGame.py:58-76); links visited entry e to it
    // kv: the parent's key as fetch_key returned it (have_kv), else it is loaded here
    __device__ u32 resolve_child(u32 node, u32 e, int a, bool *was_new, bool have_kv = false, row_t kv = 0) {
        row_t myrow; u64 rem0, rem1;
        if (have_kv) decode_key(kv, myrow, rem0, rem1); else load_key(node, myrow, rem0, rem1);
        a = uni(a);
        int i = (int)(((u32)a * p.magicW) >> 20), j = a - i * p.W;
        int iw, ih;  // the item's size: from the lanes when the kernel loaded the slot's sizes, else one dependent read
        if (have_sizes) {
            iw = (!BIG || i < 64) ? __builtin_amdgcn_readlane(w_lo, i & 63) : __builtin_amdgcn_readlane(w_hi, i - 64);
            ih = (!BIG || i < 64) ? __builtin_amdgcn_readlane(h_lo, i & 63) : __builtin_amdgcn_readlane(h_hi, i - 64);
        }
        else { iw = uni((int)wh[2 * i]); ih = uni((int)wh[2 * i + 1]); }
        myrow = apply_move_rows<row_t>(myrow, p.H, p.W, j, iw, ih);
        if (!BIG || i < 64) rem0 &= ~(1ull << (i & 63)); else rem1 &= ~(1ull << (i - 64));
        u32 child = find_or_materialize(myrow, rem0, rem1, was_new);
        if (lane_id() == 0 && child != NONE32 && e != NONE32) vis[e].child = child;
        return child;
    }
    // Ps[s][a] of a legal move from the stored float32 pi and the node's normaliser (MCTS_bpp.py:89-100)
    static __device__ __forceinline__ double prior_of(float pi, double norm, bool fallback) {
        double x = (double)pi * 1.0;
        return fallback ? (x + 1.0) / norm : x / norm;
    }
    // The visited-edge block of a node grows by doubling inside the slot's arena (old blocks are abandoned: at most the
    // live size again).  Returns false on arena overflow.
    __device__ bool grow_visited(u32 node, NodeHdr &hd) {
        u32 cap = hd.vis_cap ? 2u * hd.vis_cap : 2u, room;
        if (cap > hd.n_valid) cap = hd.n_valid;
        const u32 dst = arena_reserve(va, hd.depth, cap, &room);
        if (dst == NONE32) {
            if (lane_id() == 0) set_error(p, ERR_VIS_CAP);
            return false;
        }
        arena_commit(va, hd.depth, cap);
        for (u32 j = lane_id(); j < hd.vis_n; j += 64) {
            u32 s = hd.vis_off + j, d = dst + j;
            const uint4 *src = (const uint4 *)(vis + s);
            const uint4 a = src[0], b = src[1];
            uint4 *dst4 = (uint4 *)(vis + d);
            dst4[0] = a; dst4[1] = b;
        }
        hd.vis_off = dst; hd.vis_cap = (u16)cap;
        if (lane_id() == 0) { hdr[node].vis_off = dst; hdr[node].vis_cap = (u16)cap; }
        wave_sync();
        return true;
    }
    // The best UNVISITED legal move of a node.  Among a node's unvisited moves the PUCT score is u = (cpuct * P) * sqrt(Ns + EPS) with
    // P = float64(pi) / norm: for cpuct > 0 and probabilities pi >= 0 it is strictly increasing in the float32 pi (two different float32
    // values are >= 2^-24 apart relatively, the three float64 roundings move a value by 2^-53 each), and the reference's strict '>' in
    // ascending action order keeps the lowest index among equal scores.  So "the unvisited move that wins the argmax" is the one with
    // the largest pi, lowest index among equals -- whatever Ns is.  It is kept in the header (best_k) and changes only when that move is
    // visited for the first time: a selection then costs the node's VISITED records plus one candidate, not a pass over all its legal
    // moves (54 at 20x20 / 32, 340-460 at 50x50 / 128 where a simulation walks ~26 nodes).  `taken`: a move to treat as visited already.
    __device__ u32 rescan_best(const NodeHdr &hd, u32 taken) {
        const int lane = lane_id();
        if (vmask == nullptr) { if (lane == 0) set_error(p, ERR_PATH); return 0xFFFFu; }
        const u32 words = ((u32)hd.n_valid + 31u) >> 5;
        for (u32 w = lane; w < words; w += 64) vmask[w] = 0u;
        wave_sync();
        for (u32 j = lane; j < hd.vis_n; j += 64) { const u32 k = vis[hd.vis_off + j].idx; atomicOr(&vmask[k >> 5], 1u << (k & 31)); }
        if (lane == 0 && taken < hd.n_valid) atomicOr(&vmask[taken >> 5], 1u << (taken & 31));
        wave_sync();
        float bp = -INFINITY;
        u32 bk = 0xFFFFu;
        for (u32 k = lane; k < hd.n_valid; k += 64) {
            if ((vmask[k >> 5] >> (k & 31)) & 1u) continue;
            const float x = pPi[hd.prior_off + k];
            if (x > bp || bk == 0xFFFFu) { bp = x; bk = k; }  // k ascends within a lane: '>' keeps the lowest index among equals
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float op = __shfl_xor(bp, o);
            const u32 ok = __shfl_xor(bk, o);
            if (ok != 0xFFFFu && (bk == 0xFFFFu || op > bp || (op == bp && ok < bk))) { bp = op; bk = ok; }
        }
        wave_sync();  // the mask is reused
        return uni(bk);
    }
    // appends the visited entry of legal move k (Nsa = 0 until the backup reaches it); returns its index or NONE32.  pi = the move's
    // float32 prior (pPi[hd.prior_off + k], which the caller has in a register already).  Keeps the header's best unvisited move.
    __device__ u32 append_visited(u32 node, NodeHdr &hd, u32 k, float pi) {
        if (hd.vis_n == hd.vis_cap && !grow_visited(node, hd)) return NONE32;
        const u32 e = hd.vis_off + hd.vis_n;
        u32 nb = hd.best_k;
        if (hd.n_valid > 64u && k == hd.best_k) nb = rescan_best(hd, k);  // hd.vis_n: the records before this one; k itself counts as visited
        if (lane_id() == 0) {
            VisEntry v;
            v.q = 0.0; v.p = prior_of(pi, hd.norm, (hd.flags & HF_FALLBACK) != 0); v.n = 0u; v.child = NONE32; v.idx = k; v.pad = 0u;
            vis[e] = v;
            hdr[node].vis_n = (u16)(hd.vis_n + 1);
            if (nb != hd.best_k) hdr[node].best_k = (u16)nb;
        }
        hd.vis_n++;
        hd.best_k = (u16)nb;
        return e;
    }
    // The same selection for a node whose legal moves fit ONE pass of the wave (n_valid <= 64: nearly every node at 20x20 / 32): every
    // move is scored -- the prior run's (pi, action) pairs are requested together with the visited records, one round trip, one argmax --
    // which is cheaper there than keeping best_k current (a rescan is a dependent round trip per first visit).
    __device__ u32 select_edge_small(u32 node, NodeHdr &hd, u32 &k_out, u32 &child_out, int &act_out) {
        const int lane = lane_id();
        const double s_vis = sqrt((double)hd.ns), s_new = sqrt((double)hd.ns + 1e-8);
        double best_u = -INFINITY;
        u32 best_k = NONE32, best_e = NONE32, best_c = NONE32;
        float pi0 = 0.f;
        int act0 = 0;
        const bool mine = (u32)lane < hd.n_valid;
        if (mine) { pi0 = pPi[hd.prior_off + lane]; act0 = pAct[hd.prior_off + lane]; }
        u64 bit = 0ull;  // 1 << (legal move of this lane's visited record): OR-ed over the wave it marks the visited moves, no LDS mask
        if ((u32)lane < hd.vis_n) {  // vis_n <= n_valid <= 64: one record per lane
            const u32 e = hd.vis_off + lane;
            const VisEntry v = vis[e];
            const u32 k = v.idx, nn = v.n & NSA_MASK;
            const double cp = p.cpuct * v.p;
            best_u = nn ? v.q + cp * s_vis / (double)(1u + nn) : cp * s_new;
            best_k = k; best_e = e; best_c = v.child;
            bit = 1ull << (k & 63u);
        }
        const u64 visited = hd.vis_n ? RowOps<u64>::at(wave_scan_or(bit), 63) : 0ull;
        if (mine && !((visited >> lane) & 1ull)) {  // this lane's own legal move is unvisited: cpuct * P * sqrt(Ns + EPS)
            const double u = (p.cpuct * prior_of(pi0, hd.norm, (hd.flags & HF_FALLBACK) != 0)) * s_new;
            if (u > best_u || (u == best_u && (u32)lane < best_k)) { best_u = u; best_k = (u32)lane; best_e = NONE32; best_c = NONE32; }
        }
        double win_u = best_u;
        u32 win_k = best_k;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double ou = __shfl_xor(win_u, o);
            const u32 ok = __shfl_xor(win_k, o);
            if (ou > win_u || (ou == win_u && ok < win_k)) { win_u = ou; win_k = ok; }
        }
        k_out = win_k;
        act_out = -1;
        child_out = NONE32;
        if (win_k == NONE32) return NONE32;
        const int wl = __ffsll((long long)__ballot(best_k == win_k)) - 1;  // legal-move indices are unique among the lanes' candidates
        best_e = (u32)__builtin_amdgcn_readlane((int)best_e, wl);
        child_out = (u32)__builtin_amdgcn_readlane((int)best_c, wl);
        if (best_e == NONE32) {  // an unvisited move won: its (pi, action) sit in lane win_k
            const float pw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pi0), (int)win_k));
            act_out = __builtin_amdgcn_readlane(act0, (int)win_k);
            best_e = append_visited(node, hd, win_k, pw);
        }
        return best_e;
    }
    // PUCT argmax (MCTS_bpp.py:106-121), float64, strict '>' in ascending action order == maximum with the lowest legal-
    // move index.  Visited edges come from the node's records (Q + cpuct*P*sqrt(Ns)/(1+Nsa), or cpuct*P*sqrt(Ns+EPS) while Nsa = 0), the
    // unvisited moves are represented by the header's best_k (see rescan_best).  Returns the visited-entry index of the chosen edge
    // (appending the entry on a first visit) and its legal-move index in k_out, or NONE32.  act_out: the edge's action when it
    // was chosen among the unvisited moves (the only edges whose child still has to be resolved), else -1.
    // Memory round trips: the candidate's (pi, action) are requested with the visited records.
    __device__ u32 select_edge(u32 node, NodeHdr &hd, u32 &k_out, u32 &child_out, int &act_out) {
        if (hd.n_valid <= 64u) return select_edge_small(node, hd, k_out, child_out, act_out);
        const int lane = lane_id();
        const double s_vis = sqrt((double)hd.ns);          // math.sqrt(self.Ns[s])
        const double s_new = sqrt((double)hd.ns + 1e-8);   // math.sqrt(self.Ns[s] + EPS)
        double best_u = -INFINITY;
        u32 best_k = NONE32, best_e = NONE32, best_c = NONE32;
        const bool have_new = hd.best_k != 0xFFFFu;
        float pik = 0.f;
        int actk = 0;
        if (have_new) { pik = pPi[hd.prior_off + hd.best_k]; actk = pAct[hd.prior_off + hd.best_k]; }  // wave-uniform addresses
        for (u32 j = lane; j < hd.vis_n; j += 64) {
            const u32 e = hd.vis_off + j;
            const VisEntry v = vis[e];  // two 16-byte loads; the child link rides along: no extra round trip after the argmax
            const u32 k = v.idx;
            u32 nn = v.n & NSA_MASK, ch = v.child;
            double cp = p.cpuct * v.p;
            double u = nn ? v.q + cp * s_vis / (double)(1u + nn) : cp * s_new;
            if (u > best_u || (u == best_u && k < best_k)) { best_u = u; best_k = k; best_e = e; best_c = ch; }
        }
        // wave argmax on (u, lowest k) only; the winner's payload (entry, child) is then read from its lane
        double win_u = best_u;
        u32 win_k = best_k;
        if (hd.vis_n != 0) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double ou = __shfl_xor(win_u, o);
                const u32 ok = __shfl_xor(win_k, o);
                if (ou > win_u || (ou == win_u && ok < win_k)) { win_u = ou; win_k = ok; }
            }
        }
        if (win_k != NONE32) {  // uniform; legal-move indices are unique across lanes, so exactly one lane holds the winner
            const int wl = __ffsll((long long)__ballot(best_k == win_k)) - 1;
            best_e = (u32)__builtin_amdgcn_readlane((int)best_e, wl);
            best_c = (u32)__builtin_amdgcn_readlane((int)best_c, wl);
        }
        act_out = -1;
        if (have_new) {  // the unvisited candidate against the visited winner, same rule: larger u, lower index among equals
            const double un = (p.cpuct * prior_of(uni_f(pik), hd.norm, (hd.flags & HF_FALLBACK) != 0)) * s_new;
            if (win_k == NONE32 || un > win_u || (un == win_u && (u32)hd.best_k < win_k)) {
                win_k = hd.best_k; best_e = NONE32; best_c = NONE32; act_out = uni(actk);
            }
        }
        k_out = win_k;
        child_out = best_c;
        if (win_k == NONE32) return NONE32;
        if (best_e == NONE32) best_e = append_visited(node, hd, win_k, uni_f(pik));
        return best_e;
    }
    // visited entry of legal move k, or NONE32
    __device__ u32 find_visited(const NodeHdr &hd, u32 k) const {
        u32 found = NONE32;
        for (u32 jb = 0; jb < hd.vis_n; jb += 64) {
            u32 j = jb + lane_id();
            u64 b = __ballot(j < hd.vis_n && vis[hd.vis_off + (j < hd.vis_n ? j : 0)].idx == k);
            if (b) { found = hd.vis_off + jb + (__ffsll((long long)b) - 1); break; }
        }
        return found;
    }
    // one lane per path entry (MCTS_bpp.py:130-138)
    __device__ void backup_entry(u32 node, u32 e, double v, u32 vkind) {
        u32 nn = vis[e].n;
        u32 cnt = nn & NSA_MASK, kind = nn >> 30;
        double q = vis[e].q;
        q_update(q, kind, cnt, v, vkind);
        vis[e].q = q;
        vis[e].n = (kind << 30) | (cnt + 1u);
        hdr[node].ns += 1u;
    }
};

__device__ u64 sample_u64(u64 seed, u64 episode, u64 move) { return mix64(mix64(mix64(seed) ^ episode) ^ move); }

// CoachBPP.executeEpisode's move (CoachBPP.py:86-99) for one slot whose search budget is spent.
// action < 0: pick by p.move_rule.  Leaves phase RUNNING, EPISODE_DONE or FAILED.
template <typename row_t, bool BIG>
__device__ void play_move_impl(const DP &p, Tree<row_t, BIG> &t, int g, u32 &root, int action) {
    root = uni(root);
    NodeHdr hd = t.load_hdr(root);
    const int lane = lane_id();
    u32 chosen = NONE32;  // visited entry of the move that is played
    int free_action = -1;  // legal move played from a root the search never expanded (no statistics to keep)
    if (action >= 0) {
        u32 ksel = NONE32;
        for (u32 kb = 0; kb < hd.n_valid; kb += 64) {
            u32 k = kb + lane;
            u64 b = __ballot(k < hd.n_valid && t.pAct[hd.prior_off + (k < hd.n_valid ? k : 0)] == (u16)action);
            if (b) { ksel = kb + (__ffsll((long long)b) - 1); break; }
        }
        if (ksel != NONE32 && (hd.flags & HF_EXPANDED)) {
            chosen = t.find_visited(hd, ksel);
            if (chosen == NONE32) chosen = t.append_visited(root, hd, ksel, t.pPi[hd.prior_off + ksel]);  // a legal move the search never tried
        } else if (ksel != NONE32) {
            free_action = action;
        }
    } else if (p.move_rule == RP_MOVE_ARGMAX_FIRST) {  // most visited, lowest action among equals
        u32 best_n = 0, best_k = NONE32, best_e = NONE32;
        for (u32 j = lane; j < hd.vis_n; j += 64) {
            u32 e = hd.vis_off + j, n = t.vis[e].n & NSA_MASK, k = t.vis[e].idx;
            if (n > best_n || (n == best_n && n > 0 && k < best_k)) { best_n = n; best_k = k; best_e = e; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            u32 on = __shfl_xor(best_n, o), ok = __shfl_xor(best_k, o), oe = __shfl_xor(best_e, o);
            if (on > best_n || (on == best_n && ok < best_k)) { best_n = on; best_k = ok; best_e = oe; }
        }
        chosen = best_n ? best_e : NONE32;
    } else {  // RP_MOVE_SAMPLE: a ~ counts; inverse CDF in ascending action order over the (unordered) visited block
        u64 total = 0;
        for (u32 j = lane; j < hd.vis_n; j += 64) total += t.vis[hd.vis_off + j].n & NSA_MASK;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o);
        if (total) {
            u64 x = sample_u64(p.seed, p.episode[g], (u64)p.moves[g]);
            u64 r = __umul64hi(x, total);
            for (u32 jb = 0; jb < hd.vis_n && chosen == NONE32; jb += 64) {
                u32 j = jb + lane;
                bool act = j < hd.vis_n;
                u32 e = hd.vis_off + (act ? j : 0);
                u32 k = t.vis[e].idx;
                u64 n = act ? (u64)(t.vis[e].n & NSA_MASK) : 0ull, below = 0;
                for (u32 i = 0; i < hd.vis_n; ++i) {  // counts of all lower actions (uniform loads)
                    u32 e2 = hd.vis_off + i;
                    if (t.vis[e2].idx < k) below += t.vis[e2].n & NSA_MASK;
                }
                u64 b = __ballot(act && n > 0 && below <= r && r < below + n);
                if (b) chosen = hd.vis_off + jb + (__ffsll((long long)b) - 1);
            }
        }
    }
    if (chosen == NONE32 && free_action < 0) {
        if (lane == 0) { set_error(p, ERR_BAD_ACTION); p.phase[g] = RP_PHASE_FAILED; }
        return;
    }
    wave_sync();
    const int chosen_action = chosen != NONE32 ? (int)t.pAct[hd.prior_off + t.vis[chosen].idx] : free_action;
    u32 child = chosen != NONE32 ? t.vis[chosen].child : NONE32;
    if (child == NONE32) {
        bool was_new;
        child = t.resolve_child(root, chosen, chosen_action, &was_new);
        wave_sync();
        if (child == NONE32) { if (lane == 0) p.phase[g] = RP_PHASE_FAILED; return; }
    }
    if (p.max_examples > 0) {  // trainExamples.append([state, pi, None]) (CoachBPP.py:80)
        unsigned long long idx = 0, sp = 0;
        const u32 n_sp = p.onehot_examples ? 1u : (u32)hd.vis_n;  // hd may have grown by one entry above: the header's current view
        if (lane == 0) { idx = atomicAdd(p.ex_count, 1ull); sp = atomicAdd(p.ex_sp_cursor, (unsigned long long)n_sp); }
        idx = __shfl(idx, 0); sp = __shfl(sp, 0);
        int mv = p.moves[g];
        if ((long long)idx < p.max_examples && mv < p.N && (long long)(sp + n_sp) <= p.sp_cap) {
            const u32 *k = t.key + (size_t)root * p.KW;
            u32 *ek = p.ex_key + (size_t)idx * p.KW;
            for (int q = lane; q < p.KW; q += 64) ek[q] = k[q];
            u8 *ew = p.ex_wh + (size_t)idx * p.N * 2;
            for (int q = lane; q < 2 * p.N; q += 64) ew[q] = t.wh[q];
            if (p.onehot_examples) {
                if (lane == 0) { p.ex_sp_act[sp] = (u16)chosen_action; p.ex_sp_cnt[sp] = 1u; }
            } else {
                for (u32 q = lane; q < n_sp; q += 64) {
                    p.ex_sp_act[sp + q] = t.pAct[hd.prior_off + t.vis[hd.vis_off + q].idx];
                    p.ex_sp_cnt[sp + q] = t.vis[hd.vis_off + q].n & NSA_MASK;
                }
            }
            if (lane == 0) {
                p.ex_sp_off[idx] = (u32)sp; p.ex_sp_n[idx] = n_sp;
                p.ex_value[idx] = 0; p.ex_episode[idx] = p.episode[g]; p.ex_move[idx] = mv; p.slot_ex[(size_t)g * p.N + mv] = (u32)idx;
            }
        } else if (lane == 0) {
            set_error(p, ERR_EXAMPLES_CAP);
        }
    }
    if (p.reclaim) {  // everything at the old root's level is unreachable from now on
        arena_free_level(t.pa, hd.depth);
        arena_free_level(t.va, hd.depth);
    }
    root = child;
    t.count(CNT_MOVES);
    NodeHdr ch = t.hdr[child];
    int moves = p.moves[g] + 1;
    if (lane == 0) { p.root[g] = root; p.moves[g] = moves; p.sims_done[g] = 0; }
    if (ch.term != 0) {  // getGameEnded(next_state) != 0 (CoachBPP.py:91-99)
        row_t myrow; u64 rem0, rem1;
        t.load_key(child, myrow, rem0, rem1);
        double r;
        ranked_reward<row_t>(myrow, p.H, p.W, p.total_area[g], p.max_h[g], p.has_buf[g] != 0, p.bl[g], &r);
        t.count(CNT_EPISODES);
        if (p.max_examples > 0)  // return [(x[0], x[1], r) for x in trainExamples] (CoachBPP.py:99)
            for (int q = lane; q < moves && q < p.N; q += 64) {
                u32 idx = p.slot_ex[(size_t)g * p.N + q];
                if ((long long)idx < p.max_examples) p.ex_value[idx] = ch.term;
            }
        if (lane == 0) {
            p.last_outcome[g] = ch.term;
            p.last_score[g] = r;
            p.phase[g] = RP_PHASE_EPISODE_DONE;
            int idx = atomicAdd(p.fin_count, 1);
            if (idx < p.fin_cap) {
                p.fin_episode[idx] = p.episode[g]; p.fin_outcome[idx] = ch.term; p.fin_score[idx] = r; p.fin_moves[idx] = moves;
            } else {
                set_error(p, ERR_FINISHED_CAP);
            }
        }
    } else if (lane == 0) {
        p.last_outcome[g] = 0;
        p.last_score[g] = 0.0;
        p.phase[g] = RP_PHASE_RUNNING;
    }
}

// A slot whose episode ended takes the next instance of the pool (CoachBPP.py:123-134: the next self-play
// episode, a new MCTS with an empty tree).  Leaves the slot RUNNING at the new root, or IDLE when the pool is used up.
template <typename row_t, bool BIG>
__device__ void restart_slot_impl(const DP &p, Tree<row_t, BIG> &t, int g, u32 &root) {
    const int lane = lane_id();
    unsigned long long idx = 0;
    if (lane == 0) idx = atomicAdd(p.next_instance, 1ull);
    idx = __shfl(idx, 0);
    const PoolDesc pd = *p.pool_desc;
    if ((long long)idx >= pd.n_instances) {
        if (lane == 0) p.phase[g] = RP_PHASE_IDLE;
        return;
    }
    u8 *wh = p.item_wh + (size_t)g * p.N * 2;
    const u8 *src = pd.wh + (size_t)idx * p.N * 2;
    for (int q = lane; q < 2 * p.N; q += 64) wh[q] = src[q];
    for (int s = lane; s < p.table_cap; s += 64) t.table[s] = 0ull;
    if (lane == 0) {
        p.total_area[g] = pd.area[idx]; p.max_h[g] = pd.max_h[idx];
        p.episode[g] = pd.first_id + idx; p.moves[g] = 0; p.sims_done[g] = 0;
        p.bl[g] = *p.g_bl; p.has_buf[g] = *p.g_has_buf;
        p.last_outcome[g] = 0; p.last_score[g] = 0.0;
    }
    t.n_nodes = 0;
    t.reset_arenas();
    wave_sync();
    bool was_new;
    row_t myrow = 0;
    u64 rem0 = full_mask(p.N > 64 ? 64 : p.N), rem1 = p.N > 64 ? full_mask(p.N - 64) : 0ull;
    root = t.find_or_materialize(myrow, rem0, rem1, &was_new);
    wave_sync();
    if (lane == 0) { p.root[g] = root; p.phase[g] = root == NONE32 ? RP_PHASE_FAILED : RP_PHASE_RUNNING; }
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// MCTS.search (MCTS_bpp.py:56-139) for all slots: each wave runs simulations of its game until one
// needs the evaluator or the move's budget is spent.
#ifndef SEARCH_WAVES
#define SEARCH_WAVES 5
#endif
template <typename row_t, bool BIG>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, SEARCH_WAVES) k_search(DP p) {
    __shared__ u32 s_vmask[WAVES_PER_BLOCK][MAX_MASK_WORDS];
    extern __shared__ u16 s_stage[];  // [WAVES_PER_BLOCK][A]
    u32 *vmask = s_vmask[wave_in_block()];
    const int g = blockIdx.x * WAVES_PER_BLOCK + wave_in_block(), lane = lane_id();
    if (g >= p.G) return;
    // the slot's state is requested in one go (phase, root, simulations done, node count, item sizes): one memory round trip, not five
    int phase = p.phase[g];
    u32 root = p.root[g];
    int sims_done = p.sims_done[g];
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    Tree<row_t, BIG> t(p, g, s_stage + (size_t)wave_in_block() * p.A, s_vm[wave_in_block()], vmask);
    t.load_sizes();
    phase = uni(phase); root = uni(root); sims_done = uni(sims_done); t.n_nodes = uni(t.n_nodes);
    if (phase != RP_PHASE_RUNNING) return;  // MOVE_READY slots were handled by k_moves just before this launch
    const bool key_ahead = t.packed();  // a node's key can ride along with its header (one load instruction)
    int launched = 0;
    for (;;) {
        // A slot near the end of its game runs many evaluator-free simulations (terminal hits); the cap bounds the
        // launch's tail so one such slot cannot stall the whole wave.  Pure scheduling: results do not depend on it.
        if (p.step_cap > 0 && launched >= p.step_cap) break;
        if (sims_done >= p.sims) {  // for i in range(numMCTSSims) done (MCTS_bpp.py:37-38): the move is k_moves' job
            phase = RP_PHASE_MOVE_READY;
            break;
        }
        // ---- one simulation ----
        u32 node = root, pe0 = NONE32, pn0 = NONE32, pe1 = NONE32, pn1 = NONE32;
        int depth = 0;
        double v = 0.0;
        u32 vkind = RP_KIND_WEAK;
        bool need_eval = false, failed = false;
        for (;;) {
            // the node's key is requested WITH its header: if the chosen edge has no child yet (four simulations in five end that way)
            // the child state is built from it without another dependent round trip; otherwise the 84 bytes were a wasted line
            row_t kv = 0;
            if (key_ahead) kv = t.fetch_key(node);
            NodeHdr hd = t.load_hdr(node);
            if (hd.term != 0) {  // :81-83
                v = (double)hd.term; vkind = hdr_term_kind(hd); t.count(CNT_TERMINAL);
                break;
            }
            if (!(hd.flags & HF_EXPANDED)) { need_eval = true; break; }  // :85 leaf
            u32 ksel, child;
            int act_sel;
            const u32 vis_before = hd.vis_n;
            u32 e = t.select_edge(node, hd, ksel, child, act_sel);
            t.count(CNT_VIS_NEW, hd.vis_n - vis_before);
            if (e == NONE32 || depth >= p.N) { failed = true; break; }
            if (!BIG || depth < 64) { if (lane == depth) { pe0 = e; pn0 = node; } }
            else if (lane == depth - 64) { pe1 = e; pn1 = node; }
            depth++;
            t.count(CNT_PATH); t.count(CNT_NVALID_SEL, hd.n_valid); t.count(CNT_VIS_SEL, vis_before);
            if (child == NONE32) {  // first traversal of this edge: build the state, look it up (:125-128,:76)
                bool was_new;
                child = t.resolve_child(node, e, act_sel >= 0 ? act_sel : (int)t.pAct[hd.prior_off + ksel], &was_new, key_ahead, kv);
                if (child == NONE32) { failed = true; break; }
                if (!was_new) t.count(CNT_TRANSPOSE);
                wave_sync();
            }
            node = child;
        }
        if (failed) {
            if (lane == 0) { set_error(p, ERR_PATH); }
            phase = RP_PHASE_FAILED;
            break;
        }
        if (need_eval) {  // hand the leaf to the evaluator (nnet.predict, :87); commit kernel finishes the simulation
            u32 *pe = p.path_edge + (size_t)g * p.N, *pn = p.path_node + (size_t)g * p.N;
            if (lane < depth) { pe[lane] = pe0; pn[lane] = pn0; }
            if (BIG && lane + 64 < depth) { pe[lane + 64] = pe1; pn[lane + 64] = pn1; }
            if (lane == 0) {
                p.leaf_node[g] = node;
                p.path_len[g] = depth;
            }
            phase = RP_PHASE_WAIT_EVAL;
            break;
        }
        if (lane < depth) t.backup_entry(pn0, pe0, v, vkind);
        if (BIG && lane + 64 < depth) t.backup_entry(pn1, pe1, v, vkind);
        if (lane == 0) { p.last_v[g] = v; p.last_vkind[g] = (int)vkind; }
        sims_done++;
        launched++;
        t.count(CNT_SIMS);
        wave_sync();
    }
    t.store_sizes();
    if (lane == 0) { p.phase[g] = phase; p.sims_done[g] = sims_done; }
    t.flush_counters();
}

// CoachBPP.executeEpisode's move for every slot whose search budget is spent (RP_MOVE_ARGMAX_FIRST / RP_MOVE_SAMPLE), and the
// next instance of the pool for a slot whose episode ended.  Launched right before k_search; a separate kernel so that the
// simulation loop does not carry this code's registers.
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_moves(DP p) {
    const int g = blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (g >= p.G) return;
    if (p.phase[g] != RP_PHASE_MOVE_READY) return;
    extern __shared__ u16 s_stage[];
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    __shared__ u32 s_vmask[WAVES_PER_BLOCK][MAX_MASK_WORDS];
    Tree<row_t> t(p, g, s_stage + (size_t)wave_in_block() * p.A, s_vm[wave_in_block()], s_vmask[wave_in_block()]);
    u32 root = p.root[g];
    play_move_impl<row_t>(p, t, g, root, -1);
    wave_sync();
    if (p.phase[g] == RP_PHASE_EPISODE_DONE && p.auto_restart) {
        restart_slot_impl<row_t>(p, t, g, root);
        wave_sync();
    }
    t.store_sizes();
    t.flush_counters();
}

// np.sum(Ps[s]) over the dense A-vector in NumPy's pairwise order (MCTS_bpp.py:90,100).  mode 0:
// x[a] = float64(pi[a]) * valid[a]; mode 1 (fallback): x[a] = float64(pi[a]) * valid[a] + valid[a].
// One element of Ps[s] = pi * valids (float64 in the reference: float32 times int64).  The product is pi or (+-)0 -- exactly a
// float32 -- so the staged terms are FLOATS (round 3: half the LDS of k_commit's term buffer, 4 -> 7 workgroups per CU) and widen on
// the way out; the fallback's `+ valids` (MCTS_bpp.py:93-100) is added in float64 where the chain reads the term.
__device__ __forceinline__ float prior_term_f(const float *pi, const u32 *vmask, int a) {
    return pi[a] * (float)((vmask[a >> 5] >> (a & 31)) & 1u);  // x * 1 = x, x * 0 = (+-)0, NaN / inf * 0 = NaN: as the float64 product
}
__device__ __forceinline__ double term_at(const float *sterm, const u32 *vmask, int i, int a, int mode) {
    const double x = (double)sterm[i];
    return mode ? x + (double)((vmask[a >> 5] >> (a & 31)) & 1u) : x;
}
#define TERM_CHUNK 1024  /* 8 leaves x <= 128 elements */
__device__ double numpy_masked_sum(const DP &p, const float *pi, const u32 *vmask, double *sleaf, float *sterm, int mode) {
    const int lane = lane_id(), j = lane & 7;
    for (int lb = 0; lb < p.n_leaves; lb += 8) {
        // the 8 leaves of this pass cover one contiguous element range: build its float64 terms with coalesced loads ...
        const int lend = (lb + 8 < p.n_leaves ? lb + 8 : p.n_leaves) - 1;
        const int glo = p.leaf_lo[lb], ghi = p.leaf_lo[lend] + p.leaf_n[lend];
        for (int a = glo + lane; a < ghi; a += 64) sterm[a - glo] = prior_term_f(pi, vmask, a);
        wave_sync();
        // ... then every (leaf, accumulator) pair runs its sequential chain out of LDS
        int l = lb + (lane >> 3);
        bool act = l < p.n_leaves;
        int lo = act ? p.leaf_lo[l] - glo : 0, n = act ? p.leaf_n[l] : 0;
        double res = 0.0;
        if (n >= 8) {
            int n8 = n - (n & 7);
            double r = term_at(sterm, vmask, lo + j, glo + lo + j, mode);
            for (int i = 8 + j; i < n8; i += 8) r = r + term_at(sterm, vmask, lo + i, glo + lo + i, mode);
            double s1 = r + __shfl_down(r, 1);     // r0+r1, r2+r3, ...
            double s2 = s1 + __shfl_down(s1, 2);   // (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
            double s3 = s2 + __shfl_down(s2, 4);
            res = s3;
            if (j == 0)
                for (int i = n8; i < n; ++i) res = res + term_at(sterm, vmask, lo + i, glo + lo + i, mode);
        } else {
            if (act && j == 0) {
                res = -0.0;
                for (int i = 0; i < n; ++i) res = res + term_at(sterm, vmask, lo + i, glo + lo + i, mode);
            }
        }
        if (act && j == 0) sleaf[l] = res;
        wave_sync();
    }
    if (lane == 0)
        for (int k = 0; k < p.n_leaves - 1; ++k) sleaf[p.sched_dst[k]] = sleaf[p.sched_dst[k]] + sleaf[p.sched_src[k]];
    wave_sync();
    double total = sleaf[0];
    wave_sync();
    return total;
}

// Normaliser of one node's masked prior from the evaluator's pi (MCTS_bpp.py:88-100): np.sum(pi * valids) if that is
// positive, else the sum of the uniform fallback pi * valids + valids.  The legal moves' float32 pi go to `pi_out`.
// *best_k_out: the legal move with the largest pi, lowest index among equals (Tree::rescan_best: the first PUCT candidate of the new node).
__device__ double masked_prior(const DP &p, const float *pi, const u16 *act, float *pi_out, u32 n_valid, u32 *vmask, double *sleaf,
                               float *sterm, bool *fallback, u32 *best_k_out = nullptr) {
    const int lane = lane_id();
    const int words = (p.A + 31) >> 5;
    for (int w = lane; w < words; w += 64) vmask[w] = 0u;
    wave_sync();
    float bp = -INFINITY;
    u32 bk = 0xFFFFu;
    if (best_k_out == nullptr) {  // uniform: the common case (one-pass nodes) carries no candidate tracking
        for (u32 k = lane; k < n_valid; k += 64) { int a = act[k]; atomicOr(&vmask[a >> 5], 1u << (a & 31)); pi_out[k] = pi[a]; }
    } else {
        for (u32 k = lane; k < n_valid; k += 64) {
            int a = act[k];
            atomicOr(&vmask[a >> 5], 1u << (a & 31));
            const float x = pi[a];
            pi_out[k] = x;
            if (x > bp || bk == 0xFFFFu) { bp = x; bk = k; }
        }
    }
    if (best_k_out) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float op = __shfl_xor(bp, o);
            const u32 ok = __shfl_xor(bk, o);
            if (ok != 0xFFFFu && (bk == 0xFFFFu || op > bp || (op == bp && ok < bk))) { bp = op; bk = ok; }
        }
        *best_k_out = uni(bk);
    }
    wave_sync();
    double s = numpy_masked_sum(p, pi, vmask, sleaf, sterm, 0);
    *fallback = !(s > 0);                                             // :91
    if (*fallback) s = numpy_masked_sum(p, pi, vmask, sleaf, sterm, 1);  // :93-100  Ps = Ps + valids; Ps /= sum(Ps)
    return s;
}

// Expansion + backup for the waiting leaves (MCTS_bpp.py:87-104 then :130-139 up the path).
// LOGITS: `pi` holds the policy head's raw outputs (logits_fc, BinpackingNNet.py:69,79) and the softmax of NNet.predict
// (exp(log_softmax(x)), NNet.py:81-85) is taken here -- the row goes to LDS once, max / exp / sum / divide run in the wave, and the
// masked NumPy-order sum below reads the probabilities from LDS: the separate softmax pass over the [rows][A] matrix (one read + one
// write of 84 MB per wave of 32 768 leaves at 20x20 / 32) and the re-read by this kernel are gone.  float32 like torch.softmax:
// exp(x - max) / sum, the sum taken lane-wise then across lanes (the policy tolerance is 1e-5, not bit equality with one library).
template <typename row_t, bool LOGITS>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_commit(DP p, const float *pi, const float *vv) {
    // LDS per wave, all of it sized by the action space (commit_lds_bytes): n_leaves float64 block sums, min(A, TERM_CHUNK) float32
    // terms, (LOGITS) A float32 probabilities, ceil(A / 32) mask words -- 5.3 KB per wave at A = 640, so the workgroups per CU are
    // bounded by registers (7 waves per SIMD), not by LDS (round 2 / 3: 9.7 KB per wave with float64 terms and worst-case tables: 4)
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    const int g = blockIdx.x * WAVES_PER_BLOCK + wave_in_block(), lane = lane_id(), wv = wave_in_block();
    if (g >= p.G) return;
    const int tchunk = p.A < TERM_CHUNK ? p.A : TERM_CHUNK, mwords = ((p.A + 31) >> 5) + 1;
    double *s_leaf_w = s_dyn + (size_t)wv * p.n_leaves;
    float *s_term_w = (float *)(s_dyn + (size_t)WAVES_PER_BLOCK * p.n_leaves) + (size_t)wv * tchunk;
    float *s_soft = (float *)(s_dyn + (size_t)WAVES_PER_BLOCK * p.n_leaves) + (size_t)WAVES_PER_BLOCK * tchunk;
    u32 *s_mask_w = (u32 *)(s_soft + (LOGITS ? (size_t)WAVES_PER_BLOCK * p.A : 0)) + (size_t)wv * mwords;
    // One wave per SLOT: the slot's phase, leaf, path and evaluator row are requested together (one memory round trip) -- a wave per
    // evaluator ROW first had to fetch the row's slot (eval_slot[b]) and only then the slot's state.  Slots that wait for nothing leave.
    const int phase = p.phase[g];
    const u32 node = p.leaf_node[g];
    const int depth = p.path_len[g];
    const int b = uni(p.rows_identity ? g : p.game_row[g]);  // k_compact wrote it for exactly the slots that wait in this wave
    const u32 *pe = p.path_edge + (size_t)g * p.N, *pn = p.path_node + (size_t)g * p.N;
    u32 e0 = 0, n0 = 0, e1 = 0, n1 = 0;  // path entries of levels lane and 64 + lane (entries past `depth` are stale and unused)
    if (lane < p.N) { e0 = pe[lane]; n0 = pn[lane]; }
    if (lane + 64 < p.N) { e1 = pe[lane + 64]; n1 = pn[lane + 64]; }
    if (uni(phase) != RP_PHASE_WAIT_EVAL) return;
    const double v = (double)vv[b];  // float32 array of shape (1,) (NNet.py:85)
    const float *row = pi + (size_t)b * p.A;
    constexpr int LPL = 24;  // logits per lane: A <= 1536 (rp_commit_eval_logits checks)
    float xr[LOGITS ? LPL : 1];
    if (LOGITS) {  // the row's logits are requested with the slot's state and stay in registers (-inf past the row: exp gives 0)
#pragma unroll
        for (int i = 0; i < LPL; ++i) { const int a = lane + 64 * i; xr[i] = (64 * i < p.A && a < p.A) ? row[a] : -INFINITY; }
    }
    Tree<row_t> t(p, g);
    NodeHdr hd = t.load_hdr(uni(node));
    // backup along the stored path (:130-139) first: its loads travel with the header's
    if (lane < depth) t.backup_entry(n0, e0, v, RP_KIND_F32);
    if (lane + 64 < depth) t.backup_entry(n1, e1, v, RP_KIND_F32);
    if (LOGITS) {
        float *sp = s_soft + (size_t)wv * p.A;
        float m = xr[0];
#pragma unroll
        for (int i = 1; i < LPL; ++i) m = fmaxf(m, xr[i]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < LPL; ++i)
            if (64 * i < p.A) { xr[i] = expf(xr[i] - m); sum += xr[i]; }  // uniform: whole 64-wide groups past the row are skipped
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
#pragma unroll
        for (int i = 0; i < LPL; ++i) { const int a = lane + 64 * i; if (64 * i < p.A && a < p.A) sp[a] = xr[i] / sum; }
        wave_sync();
        row = sp;
    }
    bool fb;
    u32 bk = 0xFFFFu;  // kept for nodes whose legal moves take more than one pass of the wave (Tree::select_edge)
    double norm = masked_prior(p, row, t.pAct + hd.prior_off, t.pPi + hd.prior_off, hd.n_valid, s_mask_w, s_leaf_w, s_term_w, &fb, hd.n_valid > 64u ? &bk : nullptr);
    if (lane == 0) {  // Ps[s] (as pi + normaliser), Vs[s] = valids, Ns[s] = 0 (:89-103)
        hd.flags |= (u8)(HF_EXPANDED | (fb ? HF_FALLBACK : 0u)); hd.ns = 0; hd.norm = norm; hd.best_k = (u16)bk;
        t.hdr[node] = hd;
        p.sims_done[g] += 1; p.phase[g] = RP_PHASE_RUNNING; p.last_v[g] = v; p.last_vkind[g] = RP_KIND_F32;
        u64 *cn = p.slot_cnt + (size_t)g * CNT_N;
        cn[CNT_SIMS] += 1; cn[CNT_EXPAND] += 1; cn[CNT_NVALID_LEAF] += hd.n_valid;
    }
}

// Evaluator input of the waiting leaves: getBinItem's (N+1, H, W) planes as float32 (BinPackingGame.py:118-120,
// NNet.py:77-79): plane 0 = grid, plane i+1 = item i as ones in [0:h, 0:w] while unplaced (BinPackingGame.py:45,55).
template <typename row_t>
__device__ void write_planes(const DP &p, const u8 *wh, row_t myrow, u64 rem0, u64 rem1, float *out) {
    const int HW = p.H * p.W, lane = lane_id();
    for (int base = 0; base < HW; base += 64) {  // plane 0
        int idx = base + lane;
        int r = (int)(((u32)idx * p.magicW) >> 20), x = idx - r * p.W;
        row_t rr = __shfl(myrow, r & 63);
        if (idx < HW) out[idx] = (float)((rr >> x) & 1);
    }
    for (int i = 0; i < p.N; ++i) {
        float *pl = out + (size_t)(i + 1) * HW;
        bool unplaced = ((i < 64 ? rem0 >> i : rem1 >> (i - 64)) & 1ull) != 0;
        int w = unplaced ? wh[2 * i] : 0, h = unplaced ? wh[2 * i + 1] : 0;
        for (int idx = lane; idx < HW; idx += 64) {
            int r = (int)(((u32)idx * p.magicW) >> 20), x = idx - r * p.W;
            pl[idx] = (r < h && x < w) ? 1.0f : 0.0f;
        }
    }
}
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_leaf_planes(DP p, float *planes, long long capacity_rows) {
    const int b = blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (b >= (p.rows_identity ? p.G : *p.eval_count) || b >= capacity_rows) return;
    const int g = __builtin_amdgcn_readfirstlane(p.rows_identity ? b : p.eval_slot[b]);
    if (p.phase[g] != RP_PHASE_WAIT_EVAL) return;
    Tree<row_t> t(p, g);
    row_t myrow; u64 rem0, rem1;
    t.load_key(p.leaf_node[g], myrow, rem0, rem1);
    write_planes<row_t>(p, t.wh, myrow, rem0, rem1, planes + (size_t)b * (p.N + 1) * p.H * p.W);
}
// ------------------------------------------------------------------------------------------------
// Evaluator stem: conv_seqs[0].conv (3x3, pad 1, N+1 -> 16 channels) + max_pool2d(3, stride 2, pad 1)
// (BinpackingNNet.py:34,39-40) computed straight from the packed leaf state.
//
// The N item planes of a state are rectangles of ones anchored at the origin (BinPackingGame.py:45), so a 3x3
// convolution of plane i at pixel (r, x) only depends on WHICH taps fall inside the rectangle: the active row taps form
// an interval [lb, ub] with lb = (r == 0), ub = min(2, h - r), likewise for columns -- 5 non-empty classes per axis.
// The sum of the active taps' weights is tabulated once per weight update (T[item][row class][col class][16]); the grid
// plane goes through a 512-entry table indexed by the 3x3 bit pattern (TB).  That replaces 3.8 of the evaluator's
// 10.0 MFLOP per leaf at 20x20/32, the 52.8 KB plane write and its read-back by ~30 k table adds and a 6.4 KB write.
//
// Numerics: an output is bias + TB[pattern] + one T entry per unplaced item that reaches the pixel -- up to N + 2 terms.
// Summed in float32 in table order the partial sums wander (|sum| up to ~20 with the reference's trained 15x15 weights)
// and every add rounds at the partial sum's magnitude: 1.4e-6 off the dense convolution, which that peaked network
// amplifies to 1.6e-4 on pi.  The tables are therefore FIXED POINT: every tap sum is formed in float64 and quantised
// to an int32 with a per-channel unit q[o] = 2^-k[o], k chosen so that no reachable sum of terms can overflow
// (bound: max(P, M), P / M = bias + the largest TB entry + every item's largest entry, positive / negative parts
// separately).  Integer adds are exact and order-independent, the max-pool runs on the integers (conversion is monotone)
// and each pooled value is converted to float32 ONCE: |error| <= (N + 2) q / 2 + half an ulp of the result, with
// q ~ 2^-31 x the channel's bound (1.5e-8 for a bound of 32) -- closer to the exact sum than any float32 summation
// order, at the cost of the float32 adds it replaces (v_add_u32 for v_add_f32).
// ------------------------------------------------------------------------------------------------
#define STEM_C 16
__device__ __forceinline__ int stem_class(int r, int extent) {  // class of output coordinate r against a [0, extent) run of ones
    if (r > extent) return -1;                                    // no tap inside
    int ub = extent - r;                                          // taps 0..ub read inside the run (tap t reads coordinate r + t - 1)
    if (ub > 2) ub = 2;
    return r == 0 ? 2 + ub : ub;                                  // (lb=1: ub in {1,2} -> 3,4)  (lb=0: ub in {0,1,2} -> 0,1,2)
}
// One workgroup per output channel: float64 tap sums of the channel (N * 25 item entries, 512 grid patterns) in LDS, the
// channel's bound and unit, then the quantised tables.
__global__ void __launch_bounds__(256) k_stem_tables(int N, const float *w /*[16][N+1][3][3]*/, const float *bias, int *T, int *TB, int *bias_q, float *scale) {
    extern __shared__ double st_vals[];  // [N * 25 + 512]
    __shared__ double st_red[2][256];
    const int o = blockIdx.x, tid = threadIdx.x, nT = N * 25;
    for (int e = tid; e < nT + 512; e += 256) {
        double acc = 0.0;
        if (e < nT) {
            const int cls = e % 25, i = e / 25, rc = cls / 5, cc = cls % 5;
            const int rlb = rc >= 3 ? 1 : 0, rub = rc >= 3 ? rc - 2 : rc, clb = cc >= 3 ? 1 : 0, cub = cc >= 3 ? cc - 2 : cc;
            for (int dr = rlb; dr <= rub; ++dr)
                for (int dx = clb; dx <= cub; ++dx) acc += (double)w[((o * (N + 1) + (i + 1)) * 3 + dr) * 3 + dx];
        } else {
            const int pat = e - nT;
            for (int tap = 0; tap < 9; ++tap)
                if ((pat >> tap) & 1) acc += (double)w[(o * (N + 1) + 0) * 9 + tap];
        }
        st_vals[e] = acc;
    }
    __syncthreads();
    // bound of any reachable sum: positive and negative parts separately
    double pos = 0.0, neg = 0.0;
    for (int i = tid; i < N; i += 256) {
        double pm = 0.0, nm = 0.0;
        for (int c = 0; c < 25; ++c) { const double v = st_vals[i * 25 + c]; pm = fmax(pm, v); nm = fmax(nm, -v); }
        pos += pm; neg += nm;
    }
    double tpm = 0.0, tnm = 0.0;
    for (int q = tid; q < 512; q += 256) { const double v = st_vals[nT + q]; tpm = fmax(tpm, v); tnm = fmax(tnm, -v); }
    st_red[0][tid] = pos; st_red[1][tid] = neg;
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) { st_red[0][tid] += st_red[0][tid + sft]; st_red[1][tid] += st_red[1][tid + sft]; }
        __syncthreads();
    }
    pos = st_red[0][0]; neg = st_red[1][0];
    __syncthreads();
    st_red[0][tid] = tpm; st_red[1][tid] = tnm;
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) { st_red[0][tid] = fmax(st_red[0][tid], st_red[0][tid + sft]); st_red[1][tid] = fmax(st_red[1][tid], st_red[1][tid + sft]); }
        __syncthreads();
    }
    const double b = (double)bias[o];
    const double bound = fmax(pos + st_red[0][0] + fmax(b, 0.0), neg + st_red[1][0] + fmax(-b, 0.0));
    // k = the largest exponent with bound * 2^k + (N + 2) roundings < 2^31; at most 40 so that 2^-k stays a normal float
    int k = 40;
    if (bound > 0.0) {
        int ex;
        (void)frexp(bound, &ex);  // bound < 2^ex
        k = 30 - ex;
        if (k > 40) k = 40;
    }
    const double up = ldexp(1.0, k);
    for (int e = tid; e < nT + 512; e += 256) {
        const int q = (int)rint(st_vals[e] * up);
        if (e < nT) T[e * STEM_C + o] = q; else TB[(e - nT) * STEM_C + o] = q;
    }
    if (tid == 0) { bias_q[o] = (int)rint(b * up); scale[o] = (float)ldexp(1.0, -k); }
}

// T_LDS: the item table (N * 25 * 16 words, 51 KB at N = 32) is staged in LDS once per workgroup and the workgroup's waves
// loop over leaves -- the table rows a wave touches differ from lane to lane, which made the L1/L2 round trip of every
// 64-byte row the kernel's critical path.  Larger tables (N > 40) stay in L2.
#define STEM_NEG_INF ((int)0x80000000)  /* an output outside the image: never added to, loses every maximum */
#ifndef STEM_WAVES
#define STEM_WAVES 3  /* workgroups per CU (51 KB table each at N = 32) = waves per SIMD: the register budget (168) and the persistent grid follow it */
#endif
template <typename row_t, bool T_LDS>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK, STEM_WAVES) k_leaf_stem(DP p, float *out, float *out_relu, long long capacity_rows, int nhwc) {
    extern __shared__ __attribute__((aligned(16))) int4 sT4[];
    const int lane = lane_id();
    if (T_LDS) {
        const int n4 = (p.N * 25 + 1) * STEM_C / 4;  // + the all-zero row behind the table
        const int4 *src = (const int4 *)p.stemT;
        for (int i = threadIdx.x; i < n4; i += blockDim.x) sT4[i] = src[i];
        __syncthreads();
    }
    const long long rows = p.rows_identity ? (long long)p.G : (long long)*p.eval_count;
    const long long limit = rows < capacity_rows ? rows : capacity_rows;
    const int P = p.Hp * p.Wp;
    int bias[STEM_C];
    float unit[STEM_C];
#pragma unroll
    for (int o = 0; o < STEM_C; ++o) { bias[o] = p.stemBias[o]; unit[o] = p.stemScale[o]; }
    // T_LDS: a wave per leaf.  Table in L2 (N > 40: 204 KB at N = 128): a WORKGROUP per leaf, its four waves take every fourth pass --
    // a 50x50 leaf is 25 passes x 128 items of dependent L2 reads (0.9 ms per leaf for one wave, whatever the batch size)
    const int wv_ = wave_in_block();
    // A leaf's inputs are three dependent global round trips away (row -> slot; slot -> leaf node, phase, item sizes; node -> key) -- a
    // fifth of a wave's time per leaf when they are waited for on the spot.  The waves are persistent, so the inputs are requested
    // ahead: the slot three leaves ahead, the slot's state two ahead, the key one ahead; each is first touched a whole leaf's work
    // after its request.
    const long long stride = T_LDS ? (long long)gridDim.x * WAVES_PER_BLOCK : (long long)gridDim.x;
    auto load_slot = [&](long long bb) { return bb < limit ? (p.rows_identity ? (int)bb : p.eval_slot[bb]) : 0; };
    struct SlotIn { u32 node; int wh_lo, wh_hi, phase; };
    auto load_state = [&](int g, SlotIn &S) {  // g: wave-uniform
        S.node = p.leaf_node[g]; S.phase = p.phase[g];
        // item sizes live in lanes (item i in lane i & 63), broadcast with readlane inside the uniform item loop: no memory traffic
        const u8 *wh = p.item_wh + (size_t)g * p.N * 2;
        S.wh_lo = lane < p.N ? wh[2 * lane] | (wh[2 * lane + 1] << 8) : 0;
        S.wh_hi = lane + 64 < p.N ? wh[2 * (lane + 64)] | (wh[2 * (lane + 64) + 1] << 8) : 0;
    };
    struct KeyIn { row_t row; u32 rm[4]; };
    auto load_key_ahead = [&](int g, u32 node, KeyIn &K) {  // g, node: wave-uniform
        const u32 *k = slot_region(p, p.key, g) + (size_t)node * p.KW;
        K.row = lane < p.H ? ((const row_t *)k)[lane] : (row_t)0;
        const u32 *rwp = k + p.H * p.RW;
#pragma unroll
        for (int q = 0; q < 4; ++q) K.rm[q] = q < p.RMW ? rwp[q] : 0u;
    };
    long long b = T_LDS ? (long long)blockIdx.x * WAVES_PER_BLOCK + wv_ : (long long)blockIdx.x;
    // prologue: leaf b's chain in full, leaf b + stride up to its state, leaf b + 2 stride's slot
    int g1 = uni(load_slot(b)), g2_raw = load_slot(b + stride), g3_raw = load_slot(b + 2 * stride);
    SlotIn s1, s2;
    s1.node = 0; s1.wh_lo = s1.wh_hi = 0; s1.phase = -1; s2 = s1;
    KeyIn k1;
    k1.row = 0; k1.rm[0] = k1.rm[1] = k1.rm[2] = k1.rm[3] = 0;
    if (b < limit) { load_state(g1, s1); load_key_ahead(g1, uni(s1.node), k1); }
    int g2 = uni(g2_raw);
    if (b + stride < limit) load_state(g2, s2);
    for (; b < limit; b += stride) {
        // this leaf: everything was requested at least one leaf ago
        const int phase = uni(s1.phase);
        const u64 rem0 = (u64)uni(k1.rm[0]) | ((u64)uni(k1.rm[1]) << 32), rem1 = (u64)uni(k1.rm[2]) | ((u64)uni(k1.rm[3]) << 32);
        const row_t myrow = k1.row;
        const int wh_lo = s1.wh_lo, wh_hi = s1.wh_hi;
        // next leaf's key (its state was requested a leaf ago), the state of the one after, the slot of the third
        s1 = s2; g1 = g2;
        if (b + stride < limit) load_key_ahead(g1, uni(s1.node), k1);
        g2 = uni(g3_raw);
        if (b + 2 * stride < limit) load_state(g2, s2);
        g3_raw = load_slot(b + 3 * stride);
        if (phase != RP_PHASE_WAIT_EVAL) continue;
        float *ob = out + (size_t)b * STEM_C * P;
        // A lane owns the 2x2 block of convolution outputs (2pr + {0,1}, 2px + {0,1}) of pooled pixel (pr, px): every output is
        // computed exactly once; the 3x3/2 window's other five outputs come from the left / upper / upper-left lanes by
        // shuffle.  A pass covers 64 CONSECUTIVE pooled pixels in row-major order (lane = pixel: every lane works whatever the image
        // width -- whole rows per pass left 14 of 64 lanes idle at the 50x50 board and spent half of the rest on the halo row); passes
        // after the first start one image row + one pixel early (their first Wp + 1 lanes are the halo -- the upper and upper-left
        // neighbours of the first stored pixel: computed, not stored).
        for (int base = 0, pass = 0; base + (pass ? p.Wp + 1 : 0) < P; base += 63 - p.Wp, ++pass) {
            if (!T_LDS && (pass & (WAVES_PER_BLOCK - 1)) != wv_) continue;
            const int pq = base + lane, pr = pq / p.Wp, px = pq - pr * p.Wp;
            const bool live = pq < P;
            const int ra = 2 * pr, xa = 2 * px;
            const bool rbok = live && ra + 1 < p.H, xbok = xa + 1 < p.W;
            row_t rw[4];  // grid rows 2pr-1 .. 2pr+2 (zero outside the grid); shuffles need every lane
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int rr = ra - 1 + q;
                row_t v = __shfl(myrow, rr & 63);
                rw[q] = (live && rr >= 0 && rr < p.H) ? v : (row_t)0;
            }
            // CORNER PHASE (round 3, first pass only).  The item loop below costs 16 ds_read_b128 per item whatever the item's size, and
            // the kernel is bound by exactly that: 628 LDS instructions per 20x20 leaf at ~5 LDS cycles each on a CU's one LDS pipe
            // (profiles/r03_f_pmc_sq_own_kernels.md).  But an item is a rectangle anchored at the origin, and most are small: one whose
            // height and width are below CR x CC (8 x 8 where the image allows) reaches outputs of the top-left CR x CC pixels only.
            // Those items are summed in a layout of their own -- lane = ONE output pixel of the corner (row lane >> 3, column lane & 7),
            // 16 accumulators, 4 reads per item instead of 16 -- and the corner's sums are then handed to the lanes that own its 2x2
            // blocks (64 shuffles, about two items' worth of LDS time; taken for four small items or more).  Integer sums commute:
            // the result is bit-identical to the plain loop.
            const int zrow = p.N * 25;
            int cacc[STEM_C];
            bool cornered = false;          // uniform
            u64 small[2] = {0ull, 0ull};
            int CR = 0, CC = 0;
            if (pass == 0) {
                const int nprf = min(4, 60 / p.Wp + 1);  // pooled rows whose first four pixels sit in this pass's 64 lanes
                CR = min(8, 2 * nprf); CC = min(8, 2 * min(4, p.Wp));
                small[0] = __ballot((wh_lo >> 8) < CR && (wh_lo & 255) < CC) & rem0;
                small[1] = p.N > 64 ? (__ballot((wh_hi >> 8) < CR && (wh_hi & 255) < CC) & rem1) : 0ull;
                cornered = __popcll(small[0]) + __popcll(small[1]) >= 4;
            }
            if (cornered) {
                const int cr = lane >> 3, cx = lane & 7, r0c = cr == 0 ? 2 : 0, c0c = cx == 0 ? 2 : 0;
#pragma unroll
                for (int o = 0; o < STEM_C; ++o) cacc[o] = 0;
                for (int half = 0; half < 2; ++half)
                for (u64 m = small[half]; m; m &= m - 1) {
                    const int il = __ffsll((long long)m) - 1, i = il + 64 * half;
                    const int whi = half ? __builtin_amdgcn_readlane(wh_hi, il) : __builtin_amdgcn_readlane(wh_lo, il);
                    const int dr = (whi >> 8) - cr, dc = (whi & 255) - cx;  // >= 0: output (cr, cx) has a tap inside the rectangle
                    const int trow = (dr >= 0 && dc >= 0) ? i * 25 + (min(dr, 2) + r0c) * 5 + min(dc, 2) + c0c : zrow;
                    const int4 *tt = T_LDS ? (sT4 + trow * (STEM_C / 4)) : ((const int4 *)p.stemT + (size_t)trow * (STEM_C / 4));
#pragma unroll
                    for (int q = 0; q < 4; ++q) { int4 v = tt[q]; cacc[4 * q] += v.x; cacc[4 * q + 1] += v.y; cacc[4 * q + 2] += v.z; cacc[4 * q + 3] += v.w; }
                }
            }
            int acc[4][STEM_C];  // fixed point; outputs outside the image (odd sizes, idle lanes) start at "-inf" and stay there: the pool needs no selects
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool okj = live && ((j >> 1) ? rbok : true) && ((j & 1) ? xbok : true);
#pragma unroll
                for (int o = 0; o < STEM_C; ++o) acc[j][o] = okj ? bias[o] : STEM_NEG_INF;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // grid plane: 3x3 bit pattern, bit dr*3+dx = cell (r+dr-1, x+dx-1)
                const int x = xa + (j & 1);
                const bool ok = live && ((j >> 1) ? rbok : true) && ((j & 1) ? xbok : true);
                if (!ok) continue;
                const row_t r0 = (j >> 1) ? rw[1] : rw[0], r1 = (j >> 1) ? rw[2] : rw[1], r2 = (j >> 1) ? rw[3] : rw[2];
                u32 b0 = x ? (u32)((r0 >> (x - 1)) & 7) : (u32)((r0 << 1) & 7);
                u32 b1 = x ? (u32)((r1 >> (x - 1)) & 7) : (u32)((r1 << 1) & 7);
                u32 b2 = x ? (u32)((r2 >> (x - 1)) & 7) : (u32)((r2 << 1) & 7);
                u32 pat = b0 | (b1 << 3) | (b2 << 6);
                if (pat) {
                    const int4 *tb = (const int4 *)(p.stemTB + (size_t)pat * STEM_C);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { int4 v = tb[q]; acc[j][4 * q] += v.x; acc[j][4 * q + 1] += v.y; acc[j][4 * q + 2] += v.z; acc[j][4 * q + 3] += v.w; }
                }
            }
            // Branch-free per item: an output's class against the rectangle is min(extent - coordinate, 2) (+ 2 in row / column 0); an
            // output the rectangle does not reach reads the all-zero row behind the table instead of being skipped -- no exec-mask
            // juggling around the four outputs, and the 16 loads of an item are independent of each other.
            const int r0b = ra == 0 ? 2 : 0, c0b = xa == 0 ? 2 : 0;
            if (cornered) {  // the corner's sums go to the lanes that own its 2x2 blocks (pass 0: lane = pooled pixel pr * Wp + px)
                const bool mine = live && ra < CR && xa < CC;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int src = ((ra + (j >> 1)) * 8 + xa + (j & 1)) & 63;
                    const bool okj = mine && ((j >> 1) ? rbok : true) && ((j & 1) ? xbok : true);
#pragma unroll
                    for (int o = 0; o < STEM_C; ++o) { const int v = __shfl(cacc[o], src); acc[j][o] += okj ? v : 0; }
                }
            }
            // The items worth a look in this pass: unplaced (a placed item's plane is zero) and tall enough to reach the pass's first image
            // row -- an item's rectangle is anchored at the origin, so every pass below its height skips it without touching it (at the
            // 50x50 board 12 of 16 passes lie below nearly every item; testing all 128 items in each cost as much as the adds).
            const int thr = 2 * (base / p.Wp);
            const u64 cand[2] = {__ballot((wh_lo >> 8) >= thr) & rem0 & (cornered ? ~small[0] : ~0ull),
                                 p.N > 64 ? (__ballot((wh_hi >> 8) >= thr) & rem1 & (cornered ? ~small[1] : ~0ull)) : 0ull};
            for (int half = 0; half < 2; ++half)
            for (u64 m = cand[half]; m; m &= m - 1) {
                const int il = __ffsll((long long)m) - 1, i = il + 64 * half;
                const int whi = half ? __builtin_amdgcn_readlane(wh_hi, il) : __builtin_amdgcn_readlane(wh_lo, il);
                const int iw = whi & 255, ih = whi >> 8;
                const int dr = ih - ra, dc = iw - xa;  // >= 0: output (ra, xa) has a tap inside the rectangle
                if (__ballot(live && dr >= 0 && dc >= 0) == 0ull) continue;  // the item's rectangle (+1 border) misses every lane's block
                const int rcl[2] = {min(dr, 2) + r0b, min(dr - 1, 2)}, ccl[2] = {min(dc, 2) + c0b, min(dc - 1, 2)};
                const bool rok[2] = {live && dr >= 0, rbok && dr >= 1}, cok[2] = {dc >= 0, xbok && dc >= 1};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int trow = (rok[j >> 1] && cok[j & 1]) ? i * 25 + rcl[j >> 1] * 5 + ccl[j & 1] : zrow;
                    const int4 *tt = T_LDS ? (sT4 + trow * (STEM_C / 4)) : ((const int4 *)p.stemT + (size_t)trow * (STEM_C / 4));
#pragma unroll
                    for (int q = 0; q < 4; ++q) { int4 v = tt[q]; acc[j][4 * q] += v.x; acc[j][4 * q + 1] += v.y; acc[j][4 * q + 2] += v.z; acc[j][4 * q + 3] += v.w; }
                }
            }
            // max-pool on the integers: own block, then the right column of the left lane, the bottom row of the upper lane, the corner
            // of the upper-left; one conversion to float32 per pooled value
            const bool has_l = px > 0, has_u = lane >= p.Wp;  // the lane one image row up is in this pass (row 0 has nothing above it)
            float best[STEM_C];
#pragma unroll
            for (int o = 0; o < STEM_C; ++o) {
                const int v00 = acc[0][o], v01 = acc[1][o], v10 = acc[2][o], v11 = acc[3][o];
                const int fl = __shfl(max(v01, v11), (lane - 1) & 63);
                const int fu = __shfl(max(v10, v11), (lane - p.Wp) & 63);
                const int fc = __shfl(v11, (lane - p.Wp - 1) & 63);
                const int own = max(max(v00, v01), max(v10, v11));
                const int nb = max(max(has_l ? fl : STEM_NEG_INF, has_u ? fu : STEM_NEG_INF), (has_l && has_u) ? fc : STEM_NEG_INF);
                best[o] = (float)max(own, nb) * unit[o];
            }
            const bool store = live && (pass == 0 || lane > p.Wp);
            const int pp = pq;
            if (store && nhwc) {  // channels-last: a pixel's 16 channels are contiguous -> four 16-byte stores per lane
                float4 *o4 = (float4 *)(ob + (size_t)pp * STEM_C);
#pragma unroll
                for (int q = 0; q < 4; ++q) o4[q] = make_float4(best[4 * q], best[4 * q + 1], best[4 * q + 2], best[4 * q + 3]);
                if (out_relu) {
                    float4 *r4 = (float4 *)(out_relu + (size_t)b * STEM_C * P + (size_t)pp * STEM_C);
#pragma unroll
                    for (int q = 0; q < 4; ++q) r4[q] = make_float4(fmaxf(best[4 * q], 0.f), fmaxf(best[4 * q + 1], 0.f), fmaxf(best[4 * q + 2], 0.f), fmaxf(best[4 * q + 3], 0.f));
                }
            } else if (store) {
#pragma unroll
                for (int o = 0; o < STEM_C; ++o) ob[(size_t)o * P + pp] = best[o];
                if (out_relu) {
                    float *orl = out_relu + (size_t)b * STEM_C * P;
#pragma unroll
                    for (int o = 0; o < STEM_C; ++o) orl[(size_t)o * P + pp] = best[o] > 0.f ? best[o] : 0.f;
                }
            }
        }
    }
}

template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_leaf_states(DP p, u64 *rows_out, u8 *rem_out, int *slot_out, int max_rows) {
    const int b = blockIdx.x * WAVES_PER_BLOCK + wave_in_block(), lane = lane_id();
    if (b >= *p.eval_count || b >= max_rows) return;
    const int g = __builtin_amdgcn_readfirstlane(p.eval_slot[b]);
    if (p.phase[g] != RP_PHASE_WAIT_EVAL) return;
    Tree<row_t> t(p, g);
    row_t myrow; u64 rem0, rem1;
    t.load_key(p.leaf_node[g], myrow, rem0, rem1);
    if (lane < p.H) rows_out[(size_t)b * p.H + lane] = (u64)myrow;
    for (int i = lane; i < p.N; i += 64) rem_out[(size_t)b * p.N + i] = (u8)((i < 64 ? rem0 >> i : rem1 >> (i - 64)) & 1ull);
    if (lane == 0) slot_out[b] = g;
}

// host-provided state -> lane-resident form
template <typename row_t>
__device__ void load_host_state(const DP &p, const u64 *rows, const u8 *rem, row_t &myrow, u64 &rem0, u64 &rem1) {
    const int lane = lane_id();
    myrow = lane < p.H ? (row_t)rows[lane] : (row_t)0;
    rem0 = __ballot(lane < p.N && rem[lane] != 0);
    rem1 = __ballot(lane + 64 < p.N && rem[lane + 64 < p.N ? lane + 64 : 0] != 0);
}

// New episodes: empty tree, empty board, all items unplaced (CoachBPP.py:67-68,124); or re-rooting at a
// given state while keeping the tree (rows != nullptr).
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_set_roots(DP p, int first, int count, const u64 *rows, const u8 *rem, int clear_tree) {
    const int k = blockIdx.x * WAVES_PER_BLOCK + wave_in_block(), lane = lane_id();
    if (k >= count) return;
    const int g = first + k;
    if (clear_tree) {
        u64 *tab = slot_region(p, p.table, g);
        for (int s = lane; s < p.table_cap; s += 64) tab[s] = 0ull;
        if (lane == 0) {
            p.n_nodes[g] = 0; p.moves[g] = 0;
            p.bl[g] = *p.g_bl; p.has_buf[g] = *p.g_has_buf;  // rewards_list snapshot for this episode
            p.last_outcome[g] = 0; p.last_score[g] = 0.0;
        }
        wave_sync();
    }
    extern __shared__ u16 s_stage[];
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    Tree<row_t> t(p, g, s_stage + (size_t)wave_in_block() * p.A, s_vm[wave_in_block()]);
    if (clear_tree) { t.reset_arenas(); wave_sync(); }
    row_t myrow; u64 rem0, rem1;
    if (rows) {
        load_host_state<row_t>(p, rows + (size_t)k * p.H, rem + (size_t)k * p.N, myrow, rem0, rem1);
    } else {
        myrow = 0; rem0 = full_mask(p.N > 64 ? 64 : p.N); rem1 = p.N > 64 ? full_mask(p.N - 64) : 0ull;
    }
    bool was_new;
    u32 id = t.find_or_materialize(myrow, rem0, rem1, &was_new);
    t.store_sizes();
    t.flush_counters();
    if (lane == 0) {
        p.root[g] = id;
        p.sims_done[g] = 0;
        p.phase[g] = id == NONE32 ? RP_PHASE_FAILED : RP_PHASE_RUNNING;
    }
}

template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_advance(DP p, int first, int count, const int *action) {
    const int k = blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (k >= count) return;
    const int g = first + k;
    int phase = p.phase[g];
    if (phase != RP_PHASE_MOVE_READY && phase != RP_PHASE_RUNNING) return;
    extern __shared__ u16 s_stage[];
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    __shared__ u32 s_vmask[WAVES_PER_BLOCK][MAX_MASK_WORDS];
    Tree<row_t> t(p, g, s_stage + (size_t)wave_in_block() * p.A, s_vm[wave_in_block()], s_vmask[wave_in_block()]);
    u32 root = p.root[g];
    play_move_impl<row_t>(p, t, g, root, action[k]);
    t.store_sizes();
    t.flush_counters();
}

// counts[a] = Nsa[(root, a)] (MCTS_bpp.py:40-41)
__global__ void k_root_counts(DP p, int first, int count, u32 *out) {
    const int k = blockIdx.x * WAVES_PER_BLOCK + wave_in_block(), lane = lane_id();
    if (k >= count) return;
    const int g = first + k;
    u32 *o = out + (size_t)k * p.A;
    for (int a = lane; a < p.A; a += 64) o[a] = 0u;
    wave_sync();
    u32 root = p.root[g];
    if (root == NONE32) return;
    NodeHdr hd = slot_region(p, p.hdr, g)[root];
    const u16 *act = slot_region(p, p.pAct, g) + hd.prior_off;
    const VisEntry *ve = slot_region(p, p.vis, g) + hd.vis_off;
    for (u32 j = lane; j < hd.vis_n; j += 64) o[act[ve[j].idx]] = ve[j].n & NSA_MASK;
}

template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_pool_begin(DP p) {
    const int g = blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (g >= p.G) return;
    extern __shared__ u16 s_stage[];
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    Tree<row_t> t(p, g, s_stage + (size_t)wave_in_block() * p.A, s_vm[wave_in_block()]);
    u32 root = NONE32;
    restart_slot_impl<row_t>(p, t, g, root);
    t.store_sizes();
    t.flush_counters();
}

// Training tensors of replay examples: planes as getBinItem (BinPackingGame.py:118-120), pi = counts / sum in float64 rounded to
// float32 (MCTS_bpp.py:51-54 then torch.FloatTensor, NNet.py:46), value = ranked outcome.  The examples are PACKED -- key (rows +
// remaining words), item sizes, sparse (action, count) pairs, value -- either the context's own replay buffer (rp_examples_tensors)
// or caller-owned arrays in the same format (rp_expand_examples: a training minibatch picked by an index list out of the gathered
// replay set).  One wave per output row.
struct PackedEx {
    const u32 *key;      // [E][KW]
    const u8 *wh;        // [E][N][2]
    const int *value;    // [E]
    const long long *sp_off64;  // [E] first (action, count) pair of the example (caller-owned sets: 64-bit) ...
    const u32 *sp_off32;        // ... or the context's own 32-bit offsets
    const int *sp_n;     // [E]
    const u16 *sp_act;   // pool
    const u32 *sp_cnt;
    long long n_examples, n_sparse;  // sizes of the arrays: every index is checked against them (caller-owned data)
};
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_examples(DP p, PackedEx ex, long long first, long long count, const long long *index, float *planes,
                                                                    float *pi, float *value) {
    const long long k = (long long)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (k >= count) return;
    const long long idx = index ? index[k] : first + k;
    const int lane = lane_id();
    float *row = pi + (size_t)k * p.A;
    long long off = 0;
    int n = 0;
    bool ok = idx >= 0 && idx < ex.n_examples;
    if (ok) {
        off = ex.sp_off64 ? ex.sp_off64[idx] : (long long)ex.sp_off32[idx];
        n = ex.sp_n[idx];
        ok = off >= 0 && n >= 0 && off + n <= ex.n_sparse;
    }
    if (!ok) {  // an index outside the arrays (or a corrupt entry): a zero row and a reported error, never a stray access
        if (lane == 0) { set_error(p, ERR_BAD_EXAMPLE); value[k] = 0.f; }
        for (int a = lane; a < p.A; a += 64) row[a] = 0.f;
        float *pl = planes + (size_t)k * (p.N + 1) * p.H * p.W;
        for (int a = lane; a < (p.N + 1) * p.H * p.W; a += 64) pl[a] = 0.f;
        return;
    }
    const u32 *key = ex.key + (size_t)idx * p.KW;
    row_t myrow = lane < p.H ? ((const row_t *)key)[lane] : (row_t)0;
    const u32 *rw = key + p.H * p.RW;
    u64 rem0 = rw[0], rem1 = 0;
    if (p.RMW > 1) rem0 |= (u64)rw[1] << 32;
    if (p.RMW > 2) rem1 = rw[2];
    if (p.RMW > 3) rem1 |= (u64)rw[3] << 32;
    write_planes<row_t>(p, ex.wh + (size_t)idx * p.N * 2, myrow, rem0, rem1, planes + (size_t)k * (p.N + 1) * p.H * p.W);
    for (int a = lane; a < p.A; a += 64) row[a] = 0.f;
    u64 total = 0;
    for (int q = lane; q < n; q += 64) total += ex.sp_cnt[off + q];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) total += __shfl_xor(total, o);
    const double ds = (double)total;  // counts_sum = float(sum(counts))
    wave_sync();
    for (int q = lane; q < n; q += 64) {
        const int a = ex.sp_act[off + q];
        if (a < p.A) row[a] = (float)((double)ex.sp_cnt[off + q] / ds); else set_error(p, ERR_BAD_EXAMPLE);
    }
    if (lane == 0) value[k] = (float)ex.value[idx];
}

// ------------------------------------------------------------------------------------------------
// ItemsGenerator.items_generator(seed) on device (BinPackingGame.py:257-285): np.random.seed(seed) is MT19937's
// init_genrand, np.random.randint(lo, hi) of the legacy RandomState draws 32-bit outputs masked to the next power of two
// and rejects values above hi-1-lo (no draw at all when the range has one value).  One thread per instance; the 624-word
// state is word-major in HBM so a wave's accesses coalesce.
// ------------------------------------------------------------------------------------------------
struct Mt {
    u32 *st;      // st[i * stride]
    size_t stride;
    int pos;
    __device__ u32 &at(int i) { return st[(size_t)i * stride]; }
    __device__ void seed(u32 s) {
        for (int i = 0; i < 624; ++i) { at(i) = s; s = 1812433253u * (s ^ (s >> 30)) + (u32)i + 1u; }
        pos = 624;
    }
    __device__ void twist() {
        const u32 UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int k = 0;
        for (; k < 624 - 397; ++k) { u32 y = (at(k) & UP) | (at(k + 1) & LO); at(k) = at(k + 397) ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
        for (; k < 623; ++k) { u32 y = (at(k) & UP) | (at(k + 1) & LO); at(k) = at(k - 227) ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
        u32 y = (at(623) & UP) | (at(0) & LO);
        at(623) = at(396) ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        pos = 0;
    }
    __device__ u32 next() {
        if (pos == 624) twist();
        u32 y = at(pos++);
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    // legacy RandomState.randint(lo, hi): lo + masked rejection sample of [0, hi-1-lo]
    __device__ int randint(int lo, int hi) {
        u32 rng = (u32)(hi - 1 - lo);
        if (rng == 0) return lo;
        u32 mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        u32 v;
        while ((v = next() & mask) > rng) {}
        return lo + (int)v;
    }
};
__global__ void k_items_generator(long long n, int N, int bin_w, int bin_h, const u32 *seeds, u32 *mt_scratch, u8 *out_wh) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Mt mt; mt.st = mt_scratch + t; mt.stride = (size_t)n;
    mt.seed(seeds[t]);
    u8 L[128][4];  // [w, h, a, b]
    int len = 1;
    L[0][0] = (u8)bin_w; L[0][1] = (u8)bin_h; L[0][2] = 0; L[0][3] = 0;
    while (len < N) {
        int axis = mt.randint(0, 2);
        int idx = mt.randint(0, len);
        int w = L[idx][0], h = L[idx][1], a = L[idx][2], b = L[idx][3];
        u8 p0[4], p1[4];
        if (axis == 0) {
            if (w == 1) continue;
            int cut = mt.randint(a + 1, a + w) - a;
            p0[0] = (u8)cut; p0[1] = (u8)h; p0[2] = (u8)a; p0[3] = (u8)b;
            p1[0] = (u8)(w - cut); p1[1] = (u8)h; p1[2] = (u8)(a + cut); p1[3] = (u8)b;
        } else {
            if (h == 1) continue;
            int cut = mt.randint(b + 1, b + h) - b;
            p0[0] = (u8)w; p0[1] = (u8)cut; p0[2] = (u8)a; p0[3] = (u8)b;
            p1[0] = (u8)w; p1[1] = (u8)(h - cut); p1[2] = (u8)a; p1[3] = (u8)(b + cut);
        }
        for (int k = idx; k + 1 < len; ++k)  // item_list.pop(idx) after the two appends: the rest keeps its order
            for (int q = 0; q < 4; ++q) L[k][q] = L[k + 1][q];
        for (int q = 0; q < 4; ++q) { L[len - 1][q] = p0[q]; L[len][q] = p1[q]; }
        len++;
    }
    u8 *o = out_wh + (size_t)t * N * 2;
    for (int i = 0; i < N; ++i) { o[2 * i] = L[i][0]; o[2 * i + 1] = L[i][1]; }
}

// Compact list of the slots that wait for the evaluator, in slot order (deterministic).  One workgroup of 16 waves: wave w owns
// the contiguous segment of ceil(G / 16) slots starting at w * segment and walks it 64 slots at a time (coalesced loads, ballot +
// popcount): once to count, then -- after one barrier and the sum of the lower waves' counts -- once more to write rows in order.
__global__ void __launch_bounds__(1024) k_compact(DP p) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
    // a lane takes FOUR consecutive slots per step (one 16-byte load): 8 steps per wave at 32 768 slots instead of 32, all requested
    // before the first is used; the 4-bit results stay in registers for the second walk (which then has no loads at all)
    const int seg = (((p.G + 15) >> 4) + 255) & ~255, g0 = wid * seg, g1 = g0 + seg < p.G ? g0 + seg : p.G;
    constexpr int KEEP = 16;  // steps whose flags are kept (G <= 65 536); longer segments are read again
    u32 keep[KEEP];
    auto flags_at = [&](int base) -> u32 {
        const int g = base + 4 * lane;
        u32 m = 0;
        if (g + 3 < g1) {
            const int4 ph = *(const int4 *)(p.phase + g);
            m = (ph.x == RP_PHASE_WAIT_EVAL ? 1u : 0u) | (ph.y == RP_PHASE_WAIT_EVAL ? 2u : 0u) | (ph.z == RP_PHASE_WAIT_EVAL ? 4u : 0u) | (ph.w == RP_PHASE_WAIT_EVAL ? 8u : 0u);
        } else {
            for (int q = 0; q < 4; ++q)
                if (g + q < g1 && p.phase[g + q] == RP_PHASE_WAIT_EVAL) m |= 1u << q;
        }
        return m;
    };
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < KEEP; ++it) {
        keep[it] = g0 + it * 256 < g1 ? flags_at(g0 + it * 256) : 0u;
        cnt += __popc(keep[it]);
    }
    for (int base = g0 + KEEP * 256; base < g1; base += 256) cnt += __popc(flags_at(base));
    cnt = (int)__builtin_amdgcn_readlane((int)wave_scan_add((u32)cnt), 63);
    if (lane == 0) wsum[wid] = cnt;
    __syncthreads();
    int off = 0;
    for (int w = 0; w < wid; ++w) off += wsum[w];
    auto emit = [&](int base, u32 m) {
        const int c = __popc(m);
        const int inc = (int)wave_scan_add((u32)c);
        int row = off + inc - c;
        for (u32 mm = m; mm; mm &= mm - 1) {
            const int g = base + 4 * lane + (__ffs((int)mm) - 1);
            p.eval_slot[row] = g; p.game_row[g] = row;
            ++row;
        }
        off += __builtin_amdgcn_readlane(inc, 63);
    };
#pragma unroll
    for (int it = 0; it < KEEP; ++it)
        if (g0 + it * 256 < g1) emit(g0 + it * 256, keep[it]);
    for (int base = g0 + KEEP * 256; base < g1; base += 256) emit(base, flags_at(base));
    if (tid == 1023) *p.eval_count = off;
}
__global__ void k_reduce_counters(DP p) {  // totals over slots; one workgroup, thread k sums counter k % CNT_N over a slot stripe
    __shared__ u64 part[1024];
    const int tid = threadIdx.x, k = tid % CNT_N, stripe = tid / CNT_N, n_str = 1024 / CNT_N;
    u64 acc = 0;
    for (int g = stripe; g < p.G; g += n_str) acc += p.slot_cnt[(size_t)g * CNT_N + k];
    part[tid] = acc;
    __syncthreads();
    if (tid < CNT_N) { u64 t = 0; for (int s2 = 0; s2 < n_str; ++s2) t += part[s2 * CNT_N + tid]; p.counters[tid] = t; }
}
__global__ void k_reduce_peaks(DP p, u32 *out2) {
    __shared__ u32 m0[256], m1[256];
    u32 a = 0, b = 0;
    for (int g = threadIdx.x; g < p.G; g += 256) {  // finished episodes' marks and the running episode's chunks taken so far
        a = max(a, max(p.peak_chunks[(size_t)g * 2], p.pa_tf[(size_t)g * 2 + 1]));
        b = max(b, max(p.peak_chunks[(size_t)g * 2 + 1], p.va_tf[(size_t)g * 2 + 1]));
    }
    m0[threadIdx.x] = a; m1[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x == 0) { for (int i = 1; i < 256; ++i) { a = max(a, m0[i]); b = max(b, m1[i]); } out2[0] = a; out2[1] = b; }
}

// ---- stateless rule kernels (one wave per state) ----
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_valid_moves(DP p, long long B, const u64 *rows, const u8 *rem, const u8 *wh, u8 *mask, int *nvalid) {
    const long long b = (long long)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (b >= B) return;
    row_t myrow; u64 rem0, rem1;
    load_host_state<row_t>(p, rows + b * p.H, rem + b * p.N, myrow, rem0, rem1);
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    ValidSink sink;
    sink.act = nullptr; sink.mask = mask + b * p.A; sink.cap = 0; sink.vm = s_vm[wave_in_block()]; sink.have_sizes = 0;
    int nv = gen_valid_moves<row_t>(p, wh + b * p.N * 2, myrow, rem0, rem1, sink);
    if (nvalid && lane_id() == 0) nvalid[b] = nv;
}
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_apply_move(DP p, long long B, const u64 *rows, const u8 *rem, const u8 *wh, const int *action,
                                                                      u64 *rows_out, u8 *rem_out, int *status) {
    const long long b = (long long)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (b >= B) return;
    const int lane = lane_id();
    row_t myrow; u64 rem0, rem1;
    load_host_state<row_t>(p, rows + b * p.H, rem + b * p.N, myrow, rem0, rem1);
    int a = action[b];
    int st = 0;
    if (a < 0 || a >= p.A) {
        st = RP_ERR_ARG;
    } else {
        int i = a / p.W, j = a % p.W;  // int(action / W), int(action % W) (BinPackingGame.py:67)
        bool unplaced = ((i < 64 ? rem0 >> i : rem1 >> (i - 64)) & 1ull) != 0;
        if (!unplaced) {
            st = RP_ERR_ASSERT;  // BinPackingGame.py:69
        } else {
            const u8 *w2 = wh + b * p.N * 2;
            myrow = apply_move_rows<row_t>(myrow, p.H, p.W, j, w2[2 * i], w2[2 * i + 1]);
            if (i < 64) rem0 &= ~(1ull << i); else rem1 &= ~(1ull << (i - 64));
        }
    }
    if (lane < p.H) rows_out[b * p.H + lane] = (u64)myrow;
    for (int i = lane; i < p.N; i += 64) rem_out[b * p.N + i] = (u8)((i < 64 ? rem0 >> i : rem1 >> (i - 64)) & 1ull);
    if (lane == 0) status[b] = st;
}
template <typename row_t>
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_game_ended(DP p, long long B, const u64 *rows, const u8 *rem, const u8 *wh, const int *area,
                                                                      const int *max_h, int has_buf, double bl, int *ended, double *reward) {
    const long long b = (long long)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (b >= B) return;
    row_t myrow; u64 rem0, rem1;
    load_host_state<row_t>(p, rows + b * p.H, rem + b * p.N, myrow, rem0, rem1);
    __shared__ u64 s_vm[WAVES_PER_BLOCK][VM_WORDS];
    ValidSink sink;
    sink.act = nullptr; sink.mask = nullptr; sink.cap = 0; sink.vm = s_vm[wave_in_block()]; sink.have_sizes = 0;
    int nv = gen_valid_moves<row_t>(p, wh + b * p.N * 2, myrow, rem0, rem1, sink);
    double r = 0.0;
    int e = 0;
    if (nv == 0) e = ranked_reward<row_t>(myrow, p.H, p.W, area[b], max_h[b], has_buf != 0, bl, &r);
    if (lane_id() == 0) { ended[b] = e; reward[b] = r; }
}

__global__ void k_fill_rank(DP p, double bl, int has_buf) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < p.G) { p.bl[g] = bl; p.has_buf[g] = has_buf; }
    if (g == 0) { *p.g_bl = bl; *p.g_has_buf = has_buf; }
}

// ---- device self-tests ----
__global__ void k_selftest_sqrt(long long n, double *a, double *b) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = sqrt((double)(u32)i); b[i] = sqrt((double)(u32)i + 1e-8); }
}
__global__ void k_selftest_q(long long n, const double *q, const u8 *qk, const u32 *nsa, const double *v, const u8 *vk, double *qo, u8 *qko) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        double x = q[i]; u32 k = qk[i];
        q_update(x, k, nsa[i], v[i], vk[i]);
        qo[i] = x; qko[i] = (u8)k;
    }
}
__global__ void __launch_bounds__(64 * WAVES_PER_BLOCK) k_selftest_prior(DP p, long long B, const float *pi, const u8 *valid, double *out, u16 *act_scratch,
                                                                          float *pi_scratch) {
    __shared__ u32 s_mask[WAVES_PER_BLOCK][MAX_MASK_WORDS];
    __shared__ double s_leaf[WAVES_PER_BLOCK][MAX_LEAVES];
    __shared__ float s_term[WAVES_PER_BLOCK][TERM_CHUNK];
    const long long b = (long long)blockIdx.x * WAVES_PER_BLOCK + wave_in_block();
    if (b >= B) return;
    const int lane = lane_id(), wv = wave_in_block();
    u16 *act = act_scratch + b * p.A;
    float *pc = pi_scratch + b * p.A;
    double *o = out + b * p.A;
    int nv = 0;
    for (int base = 0; base < p.A; base += 64) {  // compact the valid actions in order
        int a = base + lane;
        bool ok = a < p.A && valid[b * p.A + a] != 0;
        u64 m = __ballot(ok);
        if (ok) act[nv + __popcll(m & lanes_below())] = (u16)a;
        if (a < p.A) o[a] = 0.0;
        nv += __popcll(m);
    }
    wave_sync();
    bool fb;
    double norm = masked_prior(p, pi + b * p.A, act, pc, (u32)nv, s_mask[wv], s_leaf[wv], s_term[wv], &fb);
    wave_sync();
    for (int k = lane; k < nv; k += 64) o[act[k]] = Tree<u32>::prior_of(pc[k], norm, fb);
}

// ------------------------------------------------------------------------------------------------
// Fused element-wise pieces of the evaluator (BinpackingNNet.py:21-27,39-40).  PyTorch-ROCm runs a convolution's bias add,
// every ReLU, the residual add and the max-pool as separate HBM-bound kernels; these do the same arithmetic in the same
// order (conv + bias, then ReLU / + skip), one pass over the activations each.  NCHW float32, contiguous.
// ------------------------------------------------------------------------------------------------
__global__ void k_nn_bias_relu(float *x, const float *bias, long long n4, int C, int HW) {  // x = relu(x + b[c])
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = ((float4 *)x)[i];
    long long e = i * 4;
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { float t = r[k] + bias[((e + k) / HW) % C]; r[k] = t > 0.f ? t : 0.f; }
    ((float4 *)x)[i] = make_float4(r[0], r[1], r[2], r[3]);
}
__global__ void k_nn_bias_residual(const float *x, const float *bias, const float *res, float *out, float *out_relu, long long n4, int C, int HW) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // out = (x + b[c]) + res ; out_relu = relu(out)
    if (i >= n4) return;
    float4 v = ((const float4 *)x)[i], s = ((const float4 *)res)[i];
    long long e = i * 4;
    float a[4] = {v.x, v.y, v.z, v.w}, b[4] = {s.x, s.y, s.z, s.w}, o[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = (a[k] + bias[((e + k) / HW) % C]) + b[k]; q[k] = o[k] > 0.f ? o[k] : 0.f; }
    ((float4 *)out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (out_relu) ((float4 *)out_relu)[i] = make_float4(q[0], q[1], q[2], q[3]);
}
// out = max_pool2d(x + b[c], 3, stride 2, pad 1); out_relu = relu(out).  One thread per output element.
__global__ void k_nn_bias_pool(const float *x, const float *bias, float *out, float *out_relu, long long n_out, int C, int H, int W, int Hp, int Wp) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    int px = (int)(i % Wp), pr = (int)((i / Wp) % Hp);
    long long plane = i / ((long long)Wp * Hp);
    const float *xp = x + plane * (long long)H * W;
    const float b = bias[plane % C];
    float m = -INFINITY;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
        int r = 2 * pr + dr;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int c = 2 * px + dx;
            if (c < 0 || c >= W) continue;
            m = fmaxf(m, xp[r * W + c] + b);
        }
    }
    out[i] = m;
    if (out_relu) out_relu[i] = m > 0.f ? m : 0.f;
}
// the same on channels-last data (x[b][r][c][ch]): one thread per output element, consecutive threads = consecutive channels
__global__ void k_nn_bias_pool_nhwc(const float *x, const float *bias, float *out, float *out_relu, long long n_out, int C, int H, int W, int Hp, int Wp) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    int ch = (int)(i % C);
    long long q = i / C;
    int px = (int)(q % Wp), pr = (int)((q / Wp) % Hp);
    long long b = q / ((long long)Wp * Hp);
    const float *xb = x + b * (long long)H * W * C + ch;
    const float bi = bias[ch];
    float m = -INFINITY;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
        int r = 2 * pr + dr;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int c = 2 * px + dx;
            if (c < 0 || c >= W) continue;
            m = fmaxf(m, xb[((long long)r * W + c) * C] + bi);
        }
    }
    out[i] = m;
    if (out_relu) out_relu[i] = m > 0.f ? m : 0.f;
}
// Same for C % 4 == 0 and < 2^32 output quads: four channels per thread (16-byte loads and stores), 32-bit index arithmetic,
// the bias added once after the maximum (x -> fl(x + b) is monotone, so max(fl(x_i + b)) == fl(max(x_i) + b) exactly).
__global__ void __launch_bounds__(256) k_nn_bias_pool_nhwc4(const float4 *__restrict__ x, const float4 *__restrict__ bias, float4 *__restrict__ out,
                                                          float4 *__restrict__ out_relu, unsigned n_quads, unsigned C4, int H, int W, int Hp, int Wp) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_quads) return;
    const unsigned ch = i % C4, q = i / C4;
    const unsigned px = q % (unsigned)Wp, q2 = q / (unsigned)Wp;
    const unsigned pr = q2 % (unsigned)Hp, b = q2 / (unsigned)Hp;
    const float4 *xb = x + (size_t)b * (unsigned)(H * W) * C4 + ch;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
        const int r = 2 * (int)pr + dr;
        if (r < 0 || r >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int c = 2 * (int)px + dx;
            if (c < 0 || c >= W) continue;
            const float4 v = xb[(unsigned)(r * W + c) * C4];
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    const float4 bi = bias[ch];
    m.x += bi.x; m.y += bi.y; m.z += bi.z; m.w += bi.w;
    out[i] = m;
    if (out_relu) out_relu[i] = make_float4(fmaxf(m.x, 0.f), fmaxf(m.y, 0.f), fmaxf(m.z, 0.f), fmaxf(m.w, 0.f));
}

// value head: out[b] = tanh(dot(z[b, :K], w) + bias)  (BinpackingNNet.py value_fc + tanh), one wave per row, 16-byte loads
__global__ void __launch_bounds__(256) k_nn_value_head(const float4 *__restrict__ z, const float4 *__restrict__ w, const float *__restrict__ bias,
                                                     float *__restrict__ out, long long B, int K4) {
    const int lane = lane_id();
    const long long b = (long long)blockIdx.x * 4 + wave_in_block();
    if (b >= B) return;
    const float4 *zr = z + b * K4;
    float acc = 0.f;
    for (int k = lane; k < K4; k += 64) {
        const float4 a = zr[k], c = w[k];
        acc += a.x * c.x; acc += a.y * c.y; acc += a.z * c.z; acc += a.w * c.w;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = tanhf(acc + bias[0]);
}

// ------------------------------------------------------------------------------------------------
// Fused residual block for the 16-channel stage (BinpackingNNet.py:15-27):
//     y = x + conv1(relu(conv0(relu(x)) + b0)) + b1 ,   3x3 convolutions, pad 1, 16 -> 16 channels, channels-last FP32
// as two implicit GEMMs on the FP32 matrix cores (v_mfma_f32_16x16x4_f32: exact f32 fma chains).  One wave per leaf:
// M = pixels (tiles of 16), N = 16 output channels, K = 9 taps x 16 input channels = 36 k-steps of 4.  The B fragments of
// both convolutions (72 VGPRs) are loaded once per wave and reused for every leaf it processes; relu(x) and the
// intermediate live in zero-bordered LDS images with a 17-float pixel stride (bank-conflict-free A-fragment reads with
// constant offsets per k-step).  Replaces 2 MIOpen convolutions + 2 element-wise kernels and their HBM round trips.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RB_STRIDE 17
#define RB_TILES 7
// fragment order of a [16][16][3][3] weight: k-step s = 4 * tap + j multiplies input channel ci = 4 * (lane >> 4) + j, so the four
// A values a lane needs for one tap are four CONSECUTIVE channels of one pixel (one 16-byte LDS read) and the four B values are
// one 16-byte load:  frag[tap][lane][j] = W[co = lane & 15][ci = 4 * (lane >> 4) + j][tap]
__global__ void k_pack_conv16(const float *w, float *frag) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 36 * 64) return;
    int j = i & 3, l = (i >> 2) & 63, tap = i >> 8;
    int co = l & 15, ci = 4 * (l >> 4) + j;
    frag[i] = w[(co * 16 + ci) * 9 + tap];
}
// One convolution over NT pixel tiles starting at tile0 (NT is a compile-time count: no predicates inside the MFMA stream).
// EPI 0: relu(acc + bias) -> padded LDS image (input of the second convolution)
// EPI 1: (acc + bias) + x -> out (and relu -> out_relu)
template <int NT, int EPI>
__device__ __forceinline__ void rb_conv_tiles(const float *img, const float (&bf)[36], float biasv, int PW, int PIX, const int *ptab, int tile0, float *img_out,
                                              const float *xl, float *ol, float *orl) {
    const int lane = lane_id();
    int abase[NT];
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int m = (tile0 + t) * 16 + (lane & 15);
        abase[t] = ptab[m < PIX ? m : 0] - (PW + 1) * RB_STRIDE + 4 * (lane >> 4);  // top-left tap of the 3x3 window, this lane's channel quad
        acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // software pipeline: the A fragments of k-step s+1 are read from LDS while the matrix cores work on step s
    float a_cur[NT], a_nxt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) a_cur[t] = img[abase[t]];
#pragma unroll
    for (int s = 0; s < 36; ++s) {
        if (s + 1 < 36) {
            const int tap = (s + 1) >> 2, dr = tap / 3, dx = tap - 3 * dr;
            const int off = (dr * PW + dx) * RB_STRIDE + ((s + 1) & 3);
#pragma unroll
            for (int t = 0; t < NT; ++t) a_nxt[t] = img[abase[t] + off];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], bf[s], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int m = (tile0 + t) * 16 + (lane >> 4) * 4 + q;
            if (m < PIX) {
                if (EPI == 0) {
                    img_out[ptab[m] + (lane & 15)] = fmaxf(acc[t][q] + biasv, 0.f);
                } else {
                    float y = (acc[t][q] + biasv) + xl[m * 16 + (lane & 15)];
                    ol[m * 16 + (lane & 15)] = y;
                    if (orl) orl[m * 16 + (lane & 15)] = fmaxf(y, 0.f);
                }
            }
        }
    }
}
template <int EPI>
__device__ __forceinline__ void rb_conv_all(const float *img, const float (&bf)[36], float biasv, int PW, int PIX, const int *ptab, int ntiles_all, float *img_out,
                                            const float *xl, float *ol, float *orl) {
    int tile0 = 0;
    for (; tile0 + RB_TILES <= ntiles_all; tile0 += RB_TILES) rb_conv_tiles<RB_TILES, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl);
    switch (ntiles_all - tile0) {  // uniform: the remainder group
        case 1: rb_conv_tiles<1, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        case 2: rb_conv_tiles<2, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        case 3: rb_conv_tiles<3, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        case 4: rb_conv_tiles<4, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        case 5: rb_conv_tiles<5, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        case 6: rb_conv_tiles<6, EPI>(img, bf, biasv, PW, PIX, ptab, tile0, img_out, xl, ol, orl); break;
        default: break;
    }
}
__global__ void __launch_bounds__(256) k_resblock16(const float *x, const float *frag0, const float *bias0, const float *frag1, const float *bias1, float *out,
                                                    float *out_relu, long long B, int S_h, int S_w) {
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    const int PW = S_w + 2, PH = S_h + 2, PIX = S_h * S_w, IMG = PH * PW * RB_STRIDE;
    int *ptab = (int *)rb_lds;                       // [PIX] padded LDS offset of every pixel's channel 0 (shared by the block)
    float *img0 = rb_lds + ((PIX + 3) & ~3) + (size_t)wv * 2 * IMG, *img1 = img0 + IMG;
    for (int i = threadIdx.x; i < PIX; i += blockDim.x) { int r = i / S_w, c = i - r * S_w; ptab[i] = ((r + 1) * PW + c + 1) * RB_STRIDE; }
    for (int i = lane; i < 2 * IMG; i += 64) img0[i] = 0.f;  // borders stay zero for the whole launch
    float bf0[36], bf1[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) { bf0[s] = frag0[((s >> 2) * 64 + lane) * 4 + (s & 3)]; bf1[s] = frag1[((s >> 2) * 64 + lane) * 4 + (s & 3)]; }
    const float bias0v = bias0[lane & 15], bias1v = bias1[lane & 15];
    const int ntiles_all = (PIX + 15) >> 4;
    __syncthreads();
    for (long long leaf = (long long)blockIdx.x * 4 + wv; leaf < B; leaf += (long long)gridDim.x * 4) {
        const float *xl = x + (size_t)leaf * PIX * 16;
        for (int e4 = lane; e4 < PIX * 4; e4 += 64) {  // relu(x) into the padded image, 16 bytes per lane
            float4 v = ((const float4 *)xl)[e4];
            float *d = img0 + ptab[e4 >> 2] + 4 * (e4 & 3);
            d[0] = fmaxf(v.x, 0.f); d[1] = fmaxf(v.y, 0.f); d[2] = fmaxf(v.z, 0.f); d[3] = fmaxf(v.w, 0.f);
        }
        lds_sync();
        rb_conv_all<0>(img0, bf0, bias0v, PW, PIX, ptab, ntiles_all, img1, nullptr, nullptr, nullptr);
        lds_sync();
        rb_conv_all<1>(img1, bf1, bias1v, PW, PIX, ptab, ntiles_all, nullptr, xl, out + (size_t)leaf * PIX * 16,
                       out_relu ? out_relu + (size_t)leaf * PIX * 16 : nullptr);
        lds_sync();
    }
}

// Both residual blocks of the 16-channel stage in one kernel (ConvSequence.res_block0 / res_block1, BinpackingNNet.py:41-46):
// four 3x3 convolutions back to back on ONE zero-bordered LDS image per wave.  All NT pixel tiles of the image keep their
// accumulators in registers, so a convolution's output overwrites its own input image once its last fragment has been read;
// the first block's output y1 stays in registers as the second block's skip operand.  Per leaf HBM sees one read of x (plus
// an L2-hot re-read in accumulator layout) and one write of the result -- the intermediate block output, its ReLU copy and the
// second kernel's launch tail are gone.
//
// The matrix product is taken TRANSPOSED: A = weights (rows = 16 output channels), B = pixels (columns = 16 pixels of a tile), so a
// lane's four accumulator values are four CONSECUTIVE channels (4 * (lane >> 4) ..) of ONE pixel (lane & 15) -- 16 contiguous bytes of
// the channels-last image.  Everything a lane touches then has that shape: x arrives as 16-byte loads in exactly this layout, an
// epilogue is bias + ReLU + one ds_write_b128 per tile, the skip operands (x, then y1) stay in registers and are added to the
// accumulators directly, the result leaves as 16-byte stores (1 KB per instruction).  With output rows = pixels the same epilogues
// needed a table lookup and a 4-byte LDS write per value, a re-read of x in accumulator layout (28 scattered 4-byte loads: 11 k cycles
// per leaf) and 4-byte output stores (13 k cycles per leaf: store-issue bound).
//
// The MFMA stream is scheduled by hand, one TAP (four k-steps) at a time: with the channel-quad fragment order a lane's pixel operands
// of a tap are one ds_read_b128 per tile (pixel stride 20 floats keeps them 16-byte aligned) and its weight operands one 16-byte load
// from L2 through a buffer resource (one address register for all 36 loads of a kernel; flat pointers make the compiler build and
// hoist a 64-bit address per load).  Weights run two taps ahead -- across convolutions too, so a convolution never starts by waiting for
// L2 -- and the pixel operands one tap ahead IN PLACE: tile t's fragment of the next tap is requested right behind the last MFMA
// that reads the current one, six MFMAs (~190 cycles) before its first use.  sched_barriers pin that order: left alone the compiler
// pairs the LDS reads of two k-steps, waits for them on the spot and issues the two dependent MFMAs back to back.
// frag = [4][9][64] float4, bias = [4][16] in execution order (b0c0, b0c1, b1c0, b1c1).
#ifndef RS_STRIDE
#define RS_STRIDE 20
#endif
#ifndef RS_LB
#define RS_LB 2
#endif
#define RS_BUF_FLAGS 0x00020000
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 rs_load_b(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
// activations stream through once: RS_STREAM_AUX = 2 marks their loads and stores non-temporal (experiment knob, default off)
#ifndef RS_STREAM_AUX
#define RS_STREAM_AUX 0
#endif
__device__ __forceinline__ f32x4 rs_load_x(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, RS_STREAM_AUX));
}
// TAIL TILE (round 3).  100 pixels of a 10x10 image are six tiles of 16 and FOUR pixels: as a seventh 16-column tile they cost a full
// tile's matrix-pipe time for a quarter of its columns (tile fill 0.893).  v_mfma_f32_4x4x1_16B_f32 multiplies sixteen independent
// 4x1 by 1x4 blocks per instruction (lane l: block l >> 2, A row / B column l & 3; result register r of lane l = A[4 (l >> 2) + r] *
// B[l], scripts/probe_mfma4x4.hip) in 8 matrix-pipe cycles instead of 32.  Block (cg, kk) = (l >> 4, (l >> 2) & 3) takes output channels
// 4 cg .. 4 cg + 3 of the four tail pixels over the input channels 4 kk .. 4 kk + 3: A = W[4 cg + (l & 3)][4 kk + q][tap] -- which is
// the ordinary fragment of lane (4 cg + (l & 3)) + 16 kk, so the same buffer serves --, B = x[pixel l & 3][4 kk + q][tap], q = 0..3:
// one ds_read_b128 and four 8-cycle MFMAs per tap (288 instead of 1 152 matrix-pipe cycles per convolution).  The four kk blocks of
// a channel group hold partial sums over their input-channel slices; they meet in the epilogue (two DPP row rotations), after which lane
// l holds channels 4 (l >> 4) .. + 3 of tail pixel l & 3 -- the accumulator layout of the ordinary tiles, so bias, ReLU, skip operands
// and the 16-byte stores are the same code.  (The tail pixels' sums associate differently from the other pixels': float32 either way.)
struct RsTail {
    int abase;     // float offset of this lane's tail pixel (l & 3): top-left tap, channel quad kk
    int voff;      // byte offset of this lane's weight fragment inside a tap's 1 KB: 16 x ((4 cg + (l & 3)) + 16 kk)
    f32x4 acc;
    f32x4 wq[3];
};
template <int CTRL> __device__ __forceinline__ float rs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ f32x4 rs_tail_sum(f32x4 v) {  // sum over the four kk blocks of a channel group (lanes l, l + 4, l + 8, l + 12 of a row)
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float s = v[k] + rs_dpp<0x128>(v[k]); r[k] = s + rs_dpp<0x124>(s); }  // row_ror:8, row_ror:4
    return r;
}
// One 3x3 convolution 16 -> 16 over NT pixel tiles (+ the tail tile).  wq: weight fragments of taps (tap, tap + 1) on entry -- taps 0 and
// 1 of the NEXT convolution (byte offset fnext) on exit.
template <int NT, bool TAIL = false>
__device__ __forceinline__ void rs_conv(const float *img, __amdgpu_buffer_rsrc_t frs, int fbase, int fnext, int PW, const int (&abase)[NT], f32x4 (&acc)[NT],
                                        f32x4 (&wq)[3], RsTail *tl = nullptr) {
    const int voff = lane_id() * 16;
    f32x4 a[NT], at = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; a[t] = *(const f32x4 *)(img + abase[t]); }
    if (TAIL) { tl->acc = (f32x4){0.f, 0.f, 0.f, 0.f}; at = *(const f32x4 *)(img + tl->abase); }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int fo = tap + 2 < 9 ? fbase + (tap + 2) * 1024 : fnext + (tap + 2 - 9) * 1024;
        wq[(tap + 2) % 3] = rs_load_b(frs, voff, fo);
        if (TAIL) tl->wq[(tap + 2) % 3] = rs_load_b(frs, tl->voff, fo);
        const f32x4 w = wq[tap % 3];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j], a[t][j], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int dr = (tap + 1) / 3, dx = (tap + 1) - 3 * dr;
        const int off = (dr * PW + dx) * RS_STRIDE;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], a[t][3], acc[t], 0, 0, 0);
            if (tap + 1 < 9) a[t] = *(const f32x4 *)(img + abase[t] + off);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAIL) {
            const f32x4 wt = tl->wq[tap % 3];
#pragma unroll
            for (int q = 0; q < 4; ++q) tl->acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wt[q], at[q], tl->acc, 0, 0, 0);
            if (tap + 1 < 9) at = *(const f32x4 *)(img + tl->abase + off);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ f32x4 rs_relu(f32x4 v) { return (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }
// Persistent waves: grid = as many workgroups as stay resident; a wave loops over tasks (IMGW leaves each).  Tables and zero borders
// are set up once; the next task's x is staged into the image right behind this task's output stores and the x after that is
// requested then, so it lands in registers during the four convolutions.  Nothing in the task loop branches on data: rows past the
// wave's pixels write to a dummy pixel, rows past the task's end read zeros and their stores are dropped by the buffer bounds check.
// TAIL: the wave's pixels are NT tiles of 16 + a tail of up to four (rs_conv).
template <int NT, bool TAIL = false>
__global__ void __launch_bounds__(256, RS_LB) k_resstage16(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                           float *__restrict__ out, float *__restrict__ out_relu, long long B, int S_h, int S_w, int IMGW,
                                                           const int *__restrict__ nrows_dev) {
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    // the wave index as a scalar: everything derived from it (task numbers, buffer resources) is then provably wave-uniform; a
    // resource the compiler takes for divergent costs a waterfall loop around every buffer load and store
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }  // only the first *nrows_dev rows hold leaves (compact rows)
    if (((long long)blockIdx.x * 4) * IMGW >= B) return;
    const int PW = S_w + 2, PH = S_h + 2, PIX = S_h * S_w, IMG = PH * PW * RS_STRIDE, MP = IMGW * PIX, WAVE_F = IMGW * IMG + RS_STRIDE;
    constexpr int NTT = NT + (TAIL ? 1 : 0);  // table rows and LDS are sized for whole tiles
    int *ptab = (int *)rb_lds;  // [16 * NTT] LDS offset (within the wave's images) of channel 0 of pixel m of the wave's IMGW leaves
    float *img = rb_lds + 16 * NTT + (size_t)wv * WAVE_F;
    for (int i = threadIdx.x; i < 16 * NTT; i += blockDim.x) {
        int im = i / PIX, pq = i - im * PIX, r = pq / S_w, c = pq - r * S_w;
        ptab[i] = i < MP ? im * IMG + ((r + 1) * PW + c + 1) * RS_STRIDE : 0;
    }
    {
        float4 *z4 = (float4 *)img;
        for (int i = lane; i < WAVE_F / 4; i += 64) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // borders (and missing leaves of the last group) stay zero
    }
    __syncthreads();
    const long long stride_leaves = (long long)gridDim.x * 4 * IMGW;
    long long leaf0 = ((long long)blockIdx.x * 4 + wv) * IMGW;
    if (leaf0 >= B) return;
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pdst[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = t * 16 + n, pc = ptab[m < MP ? m : 0];
        abase[t] = pc - (PW + 1) * RS_STRIDE + 4 * g;  // this lane's pixel of tile t: top-left tap of its 3x3 window, this lane's channel quad
        pdst[t] = (m < MP ? pc : IMGW * IMG) + 4 * g;    // where the pixel's channel quad is written (rows past the wave's pixels: the dummy pixel)
    }
    f32x4 bias4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias4[k] = *(const f32x4 *)(bias + 16 * k + 4 * g);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 4 * 36 * 64 * 4, RS_BUF_FLAGS);
    const int rowoff = n * 64 + g * 16;  // byte offset of (pixel n, channel quad g) in a tile's 1 KB of [pixel][16] floats
    auto task_bytes = [&](long long l0) { return (int)(B - l0 < IMGW ? B - l0 : IMGW) * PIX * 64; };
    f32x4 xv[NT], xs[NT], wq[3];
    // tail tile: lane = (channel group g = lane >> 4, input-channel slice kk = (lane >> 2) & 3, tail pixel lane & 3)
    RsTail tl;
    f32x4 xvt = (f32x4){0.f, 0.f, 0.f, 0.f}, xst = xvt;
    const int kk = (lane >> 2) & 3, mt = 16 * NT + (lane & 3);
    int pdst_t = 0;
    const int rowoff_t = (lane & 3) * 64 + g * 16;  // (tail pixel, channel quad g) inside the tail's 256 bytes behind the NT tiles
    if (TAIL) {
        const int pc = ptab[mt < MP ? mt : 0];
        tl.abase = pc - (PW + 1) * RS_STRIDE + 4 * kk;
        tl.voff = ((4 * g + (lane & 3)) + 16 * kk) * 16;
        pdst_t = (mt < MP ? pc : IMGW * IMG) + 4 * g;
    }
    auto load_x = [&](long long l0) {  // request a task's x in accumulator layout, zeros past its end
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)l0 * PIX * 16), 0, task_bytes(l0), RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t) xv[t] = rs_load_x(rs, rowoff, t * 1024);
        if (TAIL) xvt = rs_load_x(rs, rowoff_t, NT * 1024);
    };
    auto stage_x = [&]() {  // xs = x (skip operand of block 0); relu(x) into the padded images
#pragma unroll
        for (int t = 0; t < NT; ++t) { xs[t] = xv[t]; *(f32x4 *)(img + pdst[t]) = rs_relu(xv[t]); }
        if (TAIL) { xst = xvt; *(f32x4 *)(img + pdst_t) = rs_relu(xvt); }  // the four kk lanes of a (pixel, quad) write the same 16 bytes
    };
    load_x(leaf0);
    wq[0] = rs_load_b(frs, lane * 16, 0); wq[1] = rs_load_b(frs, lane * 16, 1024);
    if (TAIL) { tl.wq[0] = rs_load_b(frs, tl.voff, 0); tl.wq[1] = rs_load_b(frs, tl.voff, 1024); }
    stage_x();
    if (leaf0 + stride_leaves < B) load_x(leaf0 + stride_leaves);
#ifdef RS_STAMP
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define RS_T(k) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; }
#else
#define RS_T(k)
#endif
    for (; leaf0 < B; leaf0 += stride_leaves) {
        const int nbytes = task_bytes(leaf0);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)leaf0 * PIX * 16), 0, nbytes, RS_BUF_FLAGS);
#ifdef RS_STAMP
        const bool want_relu = false;  // diagnostic build: out_relu receives the stamps instead
#else
        const bool want_relu = out_relu != nullptr;
#endif
        f32x4 acc[NT];
        lds_sync();
        RS_T(1)
        rs_conv<NT, TAIL>(img, frs, 0, 9 * 1024, PW, abase, acc, wq, &tl);              // block 0, conv0
        RS_T(2)
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4 *)(img + pdst[t]) = rs_relu(acc[t] + bias4[0]);
        if (TAIL) *(f32x4 *)(img + pdst_t) = rs_relu(rs_tail_sum(tl.acc) + bias4[0]);
        lds_sync();
        RS_T(3)
        rs_conv<NT, TAIL>(img, frs, 9 * 1024, 18 * 1024, PW, abase, acc, wq, &tl);      // block 0, conv1 (+ skip x)
        RS_T(4)
#pragma unroll
        for (int t = 0; t < NT; ++t) { xs[t] = (acc[t] + bias4[1]) + xs[t]; *(f32x4 *)(img + pdst[t]) = rs_relu(xs[t]); }  // y1, kept as block 1's skip operand
        if (TAIL) { xst = (rs_tail_sum(tl.acc) + bias4[1]) + xst; *(f32x4 *)(img + pdst_t) = rs_relu(xst); }
        lds_sync();
        RS_T(5)
        rs_conv<NT, TAIL>(img, frs, 18 * 1024, 27 * 1024, PW, abase, acc, wq, &tl);     // block 1, conv0
        RS_T(6)
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4 *)(img + pdst[t]) = rs_relu(acc[t] + bias4[2]);
        if (TAIL) *(f32x4 *)(img + pdst_t) = rs_relu(rs_tail_sum(tl.acc) + bias4[2]);
        lds_sync();
        RS_T(7)
        rs_conv<NT, TAIL>(img, frs, 27 * 1024, 0, PW, abase, acc, wq, &tl);             // block 1, conv1 (+ skip y1)
        RS_T(8)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = (acc[t] + bias4[3]) + xs[t];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t]), ors, rowoff, t * 1024, RS_STREAM_AUX);
        }
        f32x4 acct = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (TAIL) {
            acct = (rs_tail_sum(tl.acc) + bias4[3]) + xst;
            if (kk == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acct), ors, rowoff_t, NT * 1024, RS_STREAM_AUX);
        }
        if (want_relu) {  // uniform: relu(result) for the next layer's input, a second set of 16-byte stores
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(out_relu + (size_t)leaf0 * PIX * 16), 0, nbytes, RS_BUF_FLAGS);
#pragma unroll
            for (int t = 0; t < NT; ++t) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t])), rrs, rowoff, t * 1024, RS_STREAM_AUX);
            if (TAIL && kk == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acct)), rrs, rowoff_t, NT * 1024, RS_STREAM_AUX);
        }
        RS_T(9)
        // the image is free (the last convolution has read it): stage the next task, request the one after
        if (leaf0 + stride_leaves < B) {
            stage_x();
            if (leaf0 + 2 * stride_leaves < B) load_x(leaf0 + 2 * stride_leaves);
        }
        RS_T(0)
    }
#undef RS_T
#ifdef RS_STAMP  // diagnostic build: per-phase shader-clock totals of all waves, added into the first 80 bytes of `out_relu`
    if (lane == 0 && out_relu)
        for (int k = 0; k < 10; ++k) atomicAdd((unsigned long long *)out_relu + k, st_acc[k]);
#endif
}

// The same recipe for the 32-channel stages on small images (5x5 and 3x3 at the 20x20 board): transposed product -- A = weights, two
// M tiles of 16 output channels, B = the 16 pixels of a tile -- so a lane's accumulators are channel quads 4 * mt + g of ONE pixel
// (two 16-byte pieces of the channels-last image), skip operands stay in registers, x / the result move as 16-byte loads / stores.
// M = the pixels of IMGW consecutive leaves (75 of 80 rows at 3 x 25), K = 9 taps x Cin channels.  Within a tap a lane group g
// multiplies input channels Cin/4 * g + 4 h + j (k-step (h, j), h < Cin / 16): four consecutive channels per half h, one
// ds_read_b128.  Fragments: frag[tap][h][mt][lane][j] = W[co = 16 mt + (lane & 15)][ci = Cin/4 * (lane >> 4) + 4 h + j][tap], 16 bytes
// per lane and (tap, h, mt), streamed from L2 through a buffer resource two half-taps ahead of their MFMAs.  Pixel operands are
// re-requested IN PLACE half a tap (16 * NT / HQ MFMAs) ahead.  The images have no padding floats (LDS: two workgroups per CU): pixel
// stride = Cin floats, a pixel's channel quad Q sits at slot Q ^ swz(pixel) with swz(p) = 3 (p >> 1) & (Cin / 4 - 1) -- without it
// sixteen pixels of a tile fall on two bank groups (16.8 LDS cycles per 16-byte read in the model of MI355X_MICROARCH.md, 9.8 with).
__global__ void k_pack_conv32(const float *w, float *frag, int Cin) {
    const int HQ = Cin / 16;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * Cin * 32) return;
    const int j = i & 3, l = (i >> 2) & 63, mt = (i >> 8) & 1, h = (i >> 9) % HQ, tap = (i >> 9) / HQ;
    const int co = 16 * mt + (l & 15), ci = (Cin / 4) * (l >> 4) + 4 * h + j;
    frag[i] = w[(co * Cin + ci) * 9 + tap];
}
// Image geometry of the 32-channel kernels (round 3): pixel stride CIN + 4 floats and ONE pad column shared by neighbouring rows
// (row pitch W + 1: the right neighbour of a row's last pixel is the left pad of the next row), + one cell behind the bottom pad row for
// the bottom-right tap of the last pixel.  A tap's operand address is then (lane's base) + (wave-uniform tap offset): ONE VALU add per
// ds_read_b128.  The XOR swizzle this replaces (quad slot Q ^ swz(pixel), images without padding floats) cost ~7 VALU instructions per
// read -- 2 286 of k_resstage32<5>'s 5 166 vector instructions -- and vector issue is what the two waves of a SIMD compete for beside
// the MFMA stream (MI355X_MICROARCH.md, 'Two waves per SIMD'): rs5 0.743 -> 0.78 of peak with the addresses alone.  Stride CIN + 4
// keeps the sixteen pixels of a tile on different bank quads (36 p mod 64 = 4 (9 p mod 16)).
#define R32_PAD 4
__host__ __device__ constexpr int r32_ps(int cin) { return cin + R32_PAD; }
__host__ __device__ inline int r32_pw(int W) { return W + 1; }
__host__ __device__ inline int r32_imgp(int H, int W) { return (H + 2) * (W + 1) + 1; }
// One 3x3 convolution CIN -> 32 over NT pixel tiles on the padded image.  abase[t]: FLOAT offset of this lane's first channel quad
// (input channels CIN/4 * g ..) of the top-left tap of tile t's pixel of this lane; wq: fragments of half-taps (0, 1) on entry, of the
// NEXT convolution's (byte offset fnext) on exit.
// Tail tile of up to four pixels for the 32-output-channel convolutions (see RsTail): block (cg, kk) = (lane >> 3, (lane >> 2) & 1)
// takes output channels 4 cg .. + 3 (eight groups) over the input-channel slice [kk CIN/2, (kk + 1) CIN/2).  Per half-tap (tap, h) the
// slice's channels (CIN/4)(2 kk + w) + 4 h + j, w = 0, 1, j = 0..3: two 16-byte operand reads, two 16-byte weight pieces -- the ordinary
// fragments of lane (4 (cg & 3) + (lane & 3)) + 16 (2 kk + w) in M tile cg >> 2 -- and eight 8-cycle MFMAs (64 matrix-pipe cycles
// against 256 for the tile).  The two kk blocks of a channel group are added in the epilogue (lanes l and l ^ 4).
// Tail state (kept as separate scalars / vectors: as one struct behind a pointer it stayed in scratch memory):
//   t_abase  float offset of the lane's tail pixel (lane & 3): top-left tap, first channel of its slice
//   t_voff0/1  byte offsets of its two weight pieces inside a half-tap's 2 KB
//   t_w0 / t_w1  [2]: the two pieces of half-taps (s, s + 1) -- one half-tap ahead is enough (a half-tap is >= 1.5 k cycles of MFMAs)
template <int NT, int CIN, int PS, bool TAIL>
__device__ __forceinline__ void r32_conv_t(const float *img, __amdgpu_buffer_rsrc_t frs, int fbase, int fnext, int PW, const int (&abase)[NT], f32x4 (&acc)[NT][2],
                                           f32x4 (&wq)[3][2], int t_abase, int t_voff0, int t_voff1, f32x4 &t_acc, f32x4 (&t_w0)[2], f32x4 (&t_w1)[2]) {
    constexpr int HQ = CIN / 16, NH = 9 * HQ;  // half-taps per tap, per convolution
    const int voff = lane_id() * 16;
    int z = 0; asm volatile("" : "+v"(z));  // opaque zero: the 9 * NT operand addresses of a convolution must not be hoisted out of the task loop
    auto a_read = [&](int t, int tap, int h) {
        const int dr = tap / 3, dx = tap - 3 * dr;
        return *(const f32x4 *)(img + abase[t] + z + ((dr * PW + dx) * PS + 4 * h));
    };
    auto at_read = [&](int tap, int h, int w) {  // tail operand: channels (CIN/4)(2 kk + w) + 4 h .. + 3 of the lane's tail pixel
        const int dr = tap / 3, dx = tap - 3 * dr;
        return *(const f32x4 *)(img + t_abase + z + ((dr * PW + dx) * PS + (CIN / 4) * w + 4 * h));
    };
    f32x4 a[NT][HQ], at[HQ][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HQ; ++h) a[t][h] = a_read(t, 0, h);
    }
    if (TAIL) {
        t_acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < HQ; ++h) { at[h][0] = at_read(0, h, 0); at[h][1] = at_read(0, h, 1); }
    }
#pragma unroll
    for (int s = 0; s < NH; ++s) {  // half-tap s = tap * HQ + h
        const int tap = s / HQ, h = s % HQ;
        {
            const int s2 = s + 2, off = s2 < NH ? fbase + s2 * 2048 : fnext + (s2 - NH) * 2048;
            wq[s2 % 3][0] = rs_load_b(frs, voff, off); wq[s2 % 3][1] = rs_load_b(frs, voff, off + 1024);
        }
        const f32x4 w0 = wq[s % 3][0], w1 = wq[s % 3][1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[t][h][j], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[t][h][j], acc[t][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[3], a[t][h][3], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[3], a[t][h][3], acc[t][1], 0, 0, 0);
            if (tap + 1 < 9) a[t][h] = a_read(t, tap + 1, h);  // this half's fragment of the next tap: first used a whole tap from here
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TAIL) {
            const f32x4 wa = t_w0[s % 2], wb = t_w1[s % 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) t_acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[j], at[h][0][j], t_acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) t_acc = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[j], at[h][1][j], t_acc, 0, 0, 0);
            asm volatile("" : "+v"(t_acc));  // keeps the chain HERE: left alone it sinks behind the convolution with all 18 operand sets alive (93 spills)
            if (tap + 1 < 9) { at[h][0] = at_read(tap + 1, h, 0); at[h][1] = at_read(tap + 1, h, 1); }
            {   // this slot is free: the pieces of half-tap s + 2 (first used a whole half-tap from here)
                const int s2 = s + 2, off = s2 < NH ? fbase + s2 * 2048 : fnext + (s2 - NH) * 2048;
                t_w0[s % 2] = rs_load_b(frs, t_voff0, off); t_w1[s % 2] = rs_load_b(frs, t_voff1, off);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int NT, int CIN, int PS = r32_ps(CIN)>
__device__ __forceinline__ void r32_conv(const float *img, __amdgpu_buffer_rsrc_t frs, int fbase, int fnext, int PW, const int (&abase)[NT], f32x4 (&acc)[NT][2],
                                         f32x4 (&wq)[3][2]) {
    f32x4 d0, d1[2], d2[2];
    r32_conv_t<NT, CIN, PS, false>(img, frs, fbase, fnext, PW, abase, acc, wq, 0, 0, 0, d0, d1, d2);
}
// frag = [4][9][2][2][64] float4, bias = [4][32] in execution order; IMGW leaves per wave, IMGW * PIX <= 16 * NT.  Persistent waves.
template <int NT>
__global__ void __launch_bounds__(256, 2) k_resstage32(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                       float *__restrict__ out, float *__restrict__ out_relu, long long B, int S_h, int S_w, int IMGW,
                                                       const int *__restrict__ nrows_dev) {
    constexpr int CIN = 32;
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }
    if (((long long)blockIdx.x * 4) * IMGW >= B) return;
    constexpr int PS = r32_ps(CIN);
    const int PW = r32_pw(S_w), PIX = S_h * S_w, IMGP = r32_imgp(S_h, S_w), MP = IMGW * PIX, WAVE_P = IMGW * IMGP + 1;  // pixels per wave incl. the dummy
    int *ptab = (int *)rb_lds;     // [16 * NT] padded pixel index (within the wave's images) of pixel m of the wave's IMGW leaves
    float *sbias = rb_lds + 16 * NT;  // [4][32]
    float *img = sbias + 128 + (size_t)wv * WAVE_P * PS;
    for (int i = threadIdx.x; i < 16 * NT; i += blockDim.x) {
        int im = i / PIX, pq = i - im * PIX, r = pq / S_w, c = pq - r * S_w;
        ptab[i] = i < MP ? im * IMGP + (r + 1) * PW + c + 1 : 0;
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) sbias[i] = bias[i];
    {
        float4 *z4 = (float4 *)img;
        for (int i = lane; i < WAVE_P * PS / 4; i += 64) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // borders (and missing leaves of the last group) stay zero
    }
    __syncthreads();
    const long long stride_leaves = (long long)gridDim.x * 4 * IMGW;
    long long leaf0 = ((long long)blockIdx.x * 4 + wv) * IMGW;
    if (leaf0 >= B) return;
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pdst[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = t * 16 + n, pc = m < MP ? ptab[m] : IMGW * IMGP;  // rows past the wave's pixels: the dummy pixel
        abase[t] = (ptab[m < MP ? m : 0] - (PW + 1)) * PS + 8 * g;  // top-left tap, input channels 8 g .. (r32_conv)
        pdst[t][0] = pc * PS + 4 * g;         // channel quad g (channels 4 g ..) and
        pdst[t][1] = pc * PS + 16 + 4 * g;    // quad 4 + g (channels 16 + 4 g ..) of this lane's pixel
    }
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 4 * 9 * CIN * 32 * 4, RS_BUF_FLAGS);
    constexpr int CONV_BYTES = 9 * CIN * 32 * 4;
    const int rowoff = n * 128 + g * 16;  // byte offset of (pixel n, channel quad g) in a tile's 2 KB of [pixel][32] floats; quad 4 + g: + 64
    auto task_bytes = [&](long long l0) { return (int)(B - l0 < IMGW ? B - l0 : IMGW) * PIX * 128; };
    f32x4 xv[NT][2], xs[NT][2], wq[3][2];
    auto load_x = [&](long long l0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)l0 * PIX * CIN), 0, task_bytes(l0), RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t) { xv[t][0] = rs_load_x(rs, rowoff, t * 2048); xv[t][1] = rs_load_x(rs, rowoff, t * 2048 + 64); }
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) { xs[t][mt] = xv[t][mt]; *(f32x4 *)(img + pdst[t][mt]) = rs_relu(xv[t][mt]); }
    };
    load_x(leaf0);
    wq[0][0] = rs_load_b(frs, lane * 16, 0); wq[0][1] = rs_load_b(frs, lane * 16, 1024);
    wq[1][0] = rs_load_b(frs, lane * 16, 2048); wq[1][1] = rs_load_b(frs, lane * 16, 3072);
    stage_x();
    if (leaf0 + stride_leaves < B) load_x(leaf0 + stride_leaves);
    for (; leaf0 < B; leaf0 += stride_leaves) {
        const int nbytes = task_bytes(leaf0);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)leaf0 * PIX * CIN), 0, nbytes, RS_BUF_FLAGS);
        f32x4 acc[NT][2];
#define R32_BIAS(k) const f32x4 ba = *(const f32x4 *)(sbias + 32 * (k) + 4 * g), bb = *(const f32x4 *)(sbias + 32 * (k) + 16 + 4 * g)
        lds_sync();
        r32_conv<NT, CIN>(img, frs, 0, CONV_BYTES, PW, abase, acc, wq);                     // block 0, conv0
        {
            R32_BIAS(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) { *(f32x4 *)(img + pdst[t][0]) = rs_relu(acc[t][0] + ba); *(f32x4 *)(img + pdst[t][1]) = rs_relu(acc[t][1] + bb); }
        }
        lds_sync();
        r32_conv<NT, CIN>(img, frs, CONV_BYTES, 2 * CONV_BYTES, PW, abase, acc, wq);        // block 0, conv1 (+ skip x)
        {
            R32_BIAS(1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                xs[t][0] = (acc[t][0] + ba) + xs[t][0]; xs[t][1] = (acc[t][1] + bb) + xs[t][1];  // y1, kept as block 1's skip operand
                *(f32x4 *)(img + pdst[t][0]) = rs_relu(xs[t][0]); *(f32x4 *)(img + pdst[t][1]) = rs_relu(xs[t][1]);
            }
        }
        lds_sync();
        r32_conv<NT, CIN>(img, frs, 2 * CONV_BYTES, 3 * CONV_BYTES, PW, abase, acc, wq);    // block 1, conv0
        {
            R32_BIAS(2);
#pragma unroll
            for (int t = 0; t < NT; ++t) { *(f32x4 *)(img + pdst[t][0]) = rs_relu(acc[t][0] + ba); *(f32x4 *)(img + pdst[t][1]) = rs_relu(acc[t][1] + bb); }
        }
        lds_sync();
        r32_conv<NT, CIN>(img, frs, 3 * CONV_BYTES, 0, PW, abase, acc, wq);                 // block 1, conv1 (+ skip y1)
        {
            R32_BIAS(3);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = (acc[t][0] + ba) + xs[t][0]; acc[t][1] = (acc[t][1] + bb) + xs[t][1];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][0]), ors, rowoff, t * 2048, RS_STREAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][1]), ors, rowoff, t * 2048 + 64, RS_STREAM_AUX);
            }
        }
#undef R32_BIAS
        if (out_relu != nullptr) {  // uniform: relu(result) for the flatten -> hidden_fc path
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(out_relu + (size_t)leaf0 * PIX * CIN), 0, nbytes, RS_BUF_FLAGS);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t][0])), rrs, rowoff, t * 2048, RS_STREAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t][1])), rrs, rowoff, t * 2048 + 64, RS_STREAM_AUX);
            }
        }
        if (leaf0 + stride_leaves < B) {
            stage_x();
            if (leaf0 + 2 * stride_leaves < B) load_x(leaf0 + 2 * stride_leaves);
        }
    }
}

// Entry of a 32-channel stage (ConvSequence.conv + max_pool2d(3, 2, 1), BinpackingNNet.py:34,39-40): 3x3 convolution
// CIN -> 32 channels over the pixels of IMGW consecutive leaves (same transposed MFMA stream as above), bias, then the 3x3 / stride-2
// max-pool.  Round 3: the padded images have a pixel stride of 36 floats for either CIN, and the convolution output + bias (32
// floats) is written IN PLACE over the interior pixels once the convolution has read them -- the pad cells are never written, so they
// stay zero for the next task and the pooling reads its windows from the same padded geometry (a neighbour outside the image reads
// the centre again).  The separate staging copy of the output (rows of 36 floats over the images) needed the whole region zeroed
// again after every task: 3.0 k of a task's 45.8 k wave cycles (scripts/probe_stage32.py with a -DCP_STAMP build).  Output: pooled x,
// channels-last, 16-byte stores.  Persistent waves; the next task's x is requested before this task's convolution.
#define CP_PS 36
#ifdef CP_STAMP
__device__ unsigned long long g_cp_stamp[8];
#endif
template <int NT, int CIN, bool TAIL = false>  // TAIL: NT tiles of 16 pixels + a tail of up to four (R32Tail)
__global__ void __launch_bounds__(256, 2) k_convpool32(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                       float *__restrict__ out, long long B, int S_h, int S_w, int IMGW, int wave_floats,
                                                       const int *__restrict__ nrows_dev) {
    constexpr int XQ = CIN / 16;  // 16-byte pieces of x per lane and tile
    constexpr int PS = CP_PS;
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }
    if (((long long)blockIdx.x * 4) * IMGW >= B) return;
    const int PW = r32_pw(S_w), PIX = S_h * S_w, IMGP = r32_imgp(S_h, S_w), MP = IMGW * PIX;
    const int Hp = (S_h + 1) >> 1, Wp = (S_w + 1) >> 1, PP = Hp * Wp;
    constexpr int NTT = NT + (TAIL ? 1 : 0);  // the tables are sized for whole tiles
    int *ptab = (int *)rb_lds;            // [16 * NTT] padded pixel index of pixel m
    int *pool = ptab + 16 * NTT;          // [IMGW * PP]: padded index of the window's centre | up << 16 | down << 17 | left << 18 | right << 19
    float *img = rb_lds + 16 * NTT + ((IMGW * PP + 3) & ~3) + (size_t)wv * wave_floats;
    for (int i = threadIdx.x; i < 16 * NTT; i += blockDim.x) {
        int im = i / PIX, pq = i - im * PIX, r = pq / S_w, c = pq - r * S_w;
        ptab[i] = i < MP ? im * IMGP + (r + 1) * PW + c + 1 : 0;
    }
    for (int i = threadIdx.x; i < IMGW * PP; i += blockDim.x) {
        int im = i / PP, pp = i - im * PP, pr = pp / Wp, px = pp - pr * Wp;
        pool[i] = (im * IMGP + (2 * pr + 1) * PW + 2 * px + 1) | ((pr > 0) << 16) | ((2 * pr + 1 < S_h) << 17) | ((px > 0) << 18) | ((2 * px + 1 < S_w) << 19);
    }
    {
        float4 *z4 = (float4 *)img;
        for (int i = lane; i < wave_floats / 4; i += 64) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // pad cells: zero once, never written again
    }
    __syncthreads();
    const long long stride_leaves = (long long)gridDim.x * 4 * IMGW;
    long long leaf0 = ((long long)blockIdx.x * 4 + wv) * IMGW;
    if (leaf0 >= B) return;
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pin[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = t * 16 + n, pc = m < MP ? ptab[m] : IMGW * IMGP;  // rows past the wave's pixels: the dummy pixel (behind the images)
        abase[t] = (ptab[m < MP ? m : 0] - (PW + 1)) * PS + CIN / 4 * g;  // top-left tap, input channels CIN/4 * g .. (r32_conv)
        pin[t] = pc * PS + 4 * g;  // this lane's pixel, channel quad g; quad 4 + g: + 16 floats (x of a 32-channel input, and the output)
    }
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 9 * CIN * 32 * 4, RS_BUF_FLAGS);
    const f32x4 ba = *(const f32x4 *)(bias + 4 * g), bb = *(const f32x4 *)(bias + 16 + 4 * g);
    const int rowoff = n * (CIN * 4) + g * 16;
    auto task_bytes = [&](long long l0) { return (int)(B - l0 < IMGW ? B - l0 : IMGW) * PIX * CIN * 4; };
    f32x4 xv[NT][XQ], wq[3][2];
    // tail tile: lane = (channel group cg = lane >> 3, input-channel slice kk = (lane >> 2) & 1, tail pixel lane & 3)
    int t_abase = 0, t_voff0 = 0, t_voff1 = 0;
    f32x4 t_acc = (f32x4){0.f, 0.f, 0.f, 0.f}, t_w0[2], t_w1[2];
    const int cg = lane >> 3, kk = (lane >> 2) & 1, mt = 16 * NT + (lane & 3);
    int pout_t = 0, pin_t = 0;
    f32x4 xvt = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int xq_t = (lane >> 2) & (CIN / 4 - 1);  // the lane's 16-byte piece of the tail pixel's x (pieces repeat across the wave: same bytes)
    const int rowoff_t = (lane & 3) * (CIN * 4) + xq_t * 16;
    if (TAIL) {
        const int pc = mt < MP ? ptab[mt] : IMGW * IMGP;
        t_abase = (ptab[mt < MP ? mt : 0] - (PW + 1)) * PS + (CIN / 2) * kk;
        const int lw = 4 * (cg & 3) + (lane & 3) + 32 * kk;
        t_voff0 = (cg >> 2) * 1024 + lw * 16; t_voff1 = (cg >> 2) * 1024 + (lw + 16) * 16;
        pout_t = pc * PS + 4 * cg;
        pin_t = pc * PS + 4 * xq_t;
    }
    auto load_x = [&](long long l0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)l0 * PIX * CIN), 0, task_bytes(l0), RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < XQ; ++q) xv[t][q] = rs_load_x(rs, rowoff, t * 16 * CIN * 4 + 64 * q);
        if (TAIL) xvt = rs_load_x(rs, rowoff_t, NT * 16 * CIN * 4);
    };
    const int cq = lane & 7;
    load_x(leaf0);
    wq[0][0] = rs_load_b(frs, lane * 16, 0); wq[0][1] = rs_load_b(frs, lane * 16, 1024);
    wq[1][0] = rs_load_b(frs, lane * 16, 2048); wq[1][1] = rs_load_b(frs, lane * 16, 3072);
    if (TAIL) {
        t_w0[0] = rs_load_b(frs, t_voff0, 0); t_w1[0] = rs_load_b(frs, t_voff1, 0);
        t_w0[1] = rs_load_b(frs, t_voff0, 2048); t_w1[1] = rs_load_b(frs, t_voff1, 2048);
    }
#ifdef CP_STAMP  // diagnostic build: per-phase shader-clock totals of all waves (rp_debug_cp_stamp)
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define CP_T(k) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; }
#else
#define CP_T(k)
#endif
    for (; leaf0 < B; leaf0 += stride_leaves) {
        const int nimg = (int)(B - leaf0 < IMGW ? B - leaf0 : IMGW);
#pragma unroll
        for (int t = 0; t < NT; ++t)  // x (no ReLU in front of a stage's first convolution) into the padded images (zeros past the task's end)
#pragma unroll
            for (int q = 0; q < XQ; ++q) *(f32x4 *)(img + pin[t] + 16 * q) = xv[t][q];
        if (TAIL) *(f32x4 *)(img + pin_t) = xvt;
        if (leaf0 + stride_leaves < B) load_x(leaf0 + stride_leaves);
        f32x4 acc[NT][2];
        lds_sync();
        CP_T(0)
        r32_conv_t<NT, CIN, PS, TAIL>(img, frs, 0, 0, PW, abase, acc, wq, t_abase, t_voff0, t_voff1, t_acc, t_w0, t_w1);
        CP_T(1)
        lds_sync();
        // convolution output + bias over the interior pixels the convolution has read (rows past the wave's pixels: the dummy pixel)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            *(f32x4 *)(img + pin[t]) = acc[t][0] + ba;
            *(f32x4 *)(img + pin[t] + 16) = acc[t][1] + bb;
        }
        if (TAIL) {  // the two input-channel slices of a channel group meet (lanes l and l ^ 4); the kk = 0 lane writes the quad
            f32x4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // bias quad cg sits in ba of lane group cg (cg < 4) or bb of lane group cg - 4
                const float b0 = __shfl(ba[k], 16 * (cg & 3)), b1 = __shfl(bb[k], 16 * (cg & 3));
                r[k] = (t_acc[k] + __shfl_xor(t_acc[k], 4)) + (cg < 4 ? b0 : b1);
            }
            if (kk == 0) *(f32x4 *)(img + pout_t) = r;
        }
        lds_sync();
        CP_T(2)
        // pooling: 8 lanes x 4 channels per pooled pixel, window geometry from the workgroup's table.  Branch-free: a neighbour outside
        // the image reads the centre again (max with itself), so all nine 16-byte reads of a window are in flight together
        float4 *o4 = (float4 *)(out + (size_t)leaf0 * PP * 32);
        for (int pq = lane >> 3; pq < nimg * PP; pq += 8) {
            const int info = pool[pq], ctr = info & 0xFFFF;
            const int up = (info >> 16) & 1, down = (info >> 17) & 1, left = (info >> 18) & 1, right = (info >> 19) & 1;
            const float4 *row = (const float4 *)(img + ctr * PS) + cq;
            const int ou = up ? -PW : 0, od = down ? PW : 0, ol = left ? -1 : 0, orr = right ? 1 : 0;
            constexpr int Q = PS / 4;
            const float4 v0 = row[0], v1 = row[ol * Q], v2 = row[orr * Q], v3 = row[ou * Q], v4 = row[(ou + ol) * Q], v5 = row[(ou + orr) * Q],
                         v6 = row[od * Q], v7 = row[(od + ol) * Q], v8 = row[(od + orr) * Q];
            float4 m;
            m.x = fmaxf(fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x)), fmaxf(fmaxf(v4.x, v5.x), fmaxf(fmaxf(v6.x, v7.x), v8.x)));
            m.y = fmaxf(fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y)), fmaxf(fmaxf(v4.y, v5.y), fmaxf(fmaxf(v6.y, v7.y), v8.y)));
            m.z = fmaxf(fmaxf(fmaxf(v0.z, v1.z), fmaxf(v2.z, v3.z)), fmaxf(fmaxf(v4.z, v5.z), fmaxf(fmaxf(v6.z, v7.z), v8.z)));
            m.w = fmaxf(fmaxf(fmaxf(v0.w, v1.w), fmaxf(v2.w, v3.w)), fmaxf(fmaxf(v4.w, v5.w), fmaxf(fmaxf(v6.w, v7.w), v8.w)));
            o4[pq * 8 + cq] = m;
        }
        CP_T(3)
        lds_sync();  // the windows have been read: the next task's x may overwrite the interior pixels
        CP_T(4)
    }
#undef CP_T
#ifdef CP_STAMP
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_cp_stamp[k], st_acc[k]);
#endif
}
#ifdef CP_STAMP
extern "C" int rp_debug_cp_stamp(unsigned long long *host8, int reset) {  // diagnostic builds only (not in include/rp_engine.h)
    if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_cp_stamp), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_cp_stamp), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// The stage kernels for images that do not fit one wave (BASELINE configs[4]: 50x50 board -> 25x25x16, 13x13x32): one WORKGROUP per
// task, the padded image(s) shared in LDS, the pixel tiles dealt to the workgroup's WAVES (4 or 8) waves in contiguous runs of NT
// (accumulators and skip operands stay in registers as above, NT per wave instead of per image; eight waves keep NT <= 5 for 40 tiles
// and put two waves on every SIMD, so one wave's barrier wait is another's MFMA time).  A convolution reads the whole image and its output
// overwrites it IN PLACE: every wave finishes its MFMA stream (barrier), then all waves write their tiles (barrier).  The MFMA
// streams, fragment orders and epilogue arithmetic are the wave kernels' (rs_conv / r32_conv): same float32 operations in the same
// order per pixel, so the results are bit-identical to them on images both can take.
// ------------------------------------------------------------------------------------------------
template <int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_resstage16_wg(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                          float *__restrict__ out, float *__restrict__ out_relu, long long B, int S_h, int S_w,
                                                          const int *__restrict__ nrows_dev) {
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }
    if ((long long)blockIdx.x >= B) return;
    const int PW = S_w + 2, PH = S_h + 2, PIX = S_h * S_w, IMG = PH * PW * RS_STRIDE;
    float *img = rb_lds;  // [IMG] + one dummy pixel
    {
        float4 *z4 = (float4 *)img;
        for (int i = threadIdx.x; i < (IMG + RS_STRIDE) / 4; i += 64 * WAVES) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // borders stay zero
    }
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pdst[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = (wv * NT + t) * 16 + n, mm = m < PIX ? m : 0, r = mm / S_w, c = mm - r * S_w, pc = ((r + 1) * PW + c + 1) * RS_STRIDE;
        abase[t] = pc - (PW + 1) * RS_STRIDE + 4 * g;
        pdst[t] = (m < PIX ? pc : IMG) + 4 * g;  // rows past the image: the dummy pixel
    }
    f32x4 bias4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bias4[k] = *(const f32x4 *)(bias + 16 * k + 4 * g);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 4 * 36 * 64 * 4, RS_BUF_FLAGS);
    const int rowoff = n * 64 + g * 16, tile0 = wv * NT;
    f32x4 xs[NT], wq[3];
    wq[0] = rs_load_b(frs, lane * 16, 0); wq[1] = rs_load_b(frs, lane * 16, 1024);
    lds_barrier();
    for (long long leaf = blockIdx.x; leaf < B; leaf += gridDim.x) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)leaf * PIX * 16), 0, PIX * 64, RS_BUF_FLAGS);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)leaf * PIX * 16), 0, PIX * 64, RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t) { xs[t] = rs_load_x(xrs, rowoff, (tile0 + t) * 1024); *(f32x4 *)(img + pdst[t]) = rs_relu(xs[t]); }  // zeros past the image
        f32x4 acc[NT];
        lds_barrier();
        rs_conv<NT>(img, frs, 0, 9 * 1024, PW, abase, acc, wq);              // block 0, conv0
        lds_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4 *)(img + pdst[t]) = rs_relu(acc[t] + bias4[0]);
        lds_barrier();
        rs_conv<NT>(img, frs, 9 * 1024, 18 * 1024, PW, abase, acc, wq);      // block 0, conv1 (+ skip x)
        lds_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t) { xs[t] = (acc[t] + bias4[1]) + xs[t]; *(f32x4 *)(img + pdst[t]) = rs_relu(xs[t]); }
        lds_barrier();
        rs_conv<NT>(img, frs, 18 * 1024, 27 * 1024, PW, abase, acc, wq);     // block 1, conv0
        lds_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t) *(f32x4 *)(img + pdst[t]) = rs_relu(acc[t] + bias4[2]);
        lds_barrier();
        rs_conv<NT>(img, frs, 27 * 1024, 0, PW, abase, acc, wq);             // block 1, conv1 (+ skip y1)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = (acc[t] + bias4[3]) + xs[t];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t]), ors, rowoff, (tile0 + t) * 1024, RS_STREAM_AUX);  // dropped past the image
        }
        if (out_relu != nullptr) {
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(out_relu + (size_t)leaf * PIX * 16), 0, PIX * 64, RS_BUF_FLAGS);
#pragma unroll
            for (int t = 0; t < NT; ++t) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t])), rrs, rowoff, (tile0 + t) * 1024, RS_STREAM_AUX);
        }
        lds_barrier();  // every wave has read the image: the next task may overwrite it
    }
}

// 32-channel stage, IMGW leaves per workgroup (13x13: three leaves = 507 of 512 tile rows).  Swizzled image without padding floats.
template <int NT, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_resstage32_wg(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                          float *__restrict__ out, float *__restrict__ out_relu, long long B, int S_h, int S_w, int IMGW,
                                                          const int *__restrict__ nrows_dev) {
    constexpr int CIN = 32;
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }
    if ((long long)blockIdx.x * IMGW >= B) return;
    constexpr int PS = r32_ps(CIN);
    const int PW = r32_pw(S_w), PIX = S_h * S_w, IMGP = r32_imgp(S_h, S_w), MP = IMGW * PIX, WG_P = IMGW * IMGP + 1;  // pixels incl. the dummy
    float *sbias = rb_lds;  // [4][32]
    float *img = sbias + 128;
    for (int i = threadIdx.x; i < 128; i += 64 * WAVES) sbias[i] = bias[i];
    {
        float4 *z4 = (float4 *)img;
        for (int i = threadIdx.x; i < WG_P * PS / 4; i += 64 * WAVES) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pdst[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = (wv * NT + t) * 16 + n, mm = m < MP ? m : 0, im = mm / PIX, pq = mm - im * PIX, r = pq / S_w, c = pq - r * S_w;
        const int pix = im * IMGP + (r + 1) * PW + c + 1, pc = m < MP ? pix : IMGW * IMGP;
        abase[t] = (pix - (PW + 1)) * PS + 8 * g;
        pdst[t][0] = pc * PS + 4 * g;
        pdst[t][1] = pc * PS + 16 + 4 * g;
    }
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 4 * 9 * CIN * 32 * 4, RS_BUF_FLAGS);
    constexpr int CONV_BYTES = 9 * CIN * 32 * 4;
    const int rowoff = n * 128 + g * 16, tile0 = wv * NT;
    f32x4 xs[NT][2], wq[3][2];
    wq[0][0] = rs_load_b(frs, lane * 16, 0); wq[0][1] = rs_load_b(frs, lane * 16, 1024);
    wq[1][0] = rs_load_b(frs, lane * 16, 2048); wq[1][1] = rs_load_b(frs, lane * 16, 3072);
    lds_barrier();
    const long long stride_leaves = (long long)gridDim.x * IMGW;
    for (long long leaf0 = (long long)blockIdx.x * IMGW; leaf0 < B; leaf0 += stride_leaves) {
        const int nbytes = (int)(B - leaf0 < IMGW ? B - leaf0 : IMGW) * PIX * 128;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)leaf0 * PIX * CIN), 0, nbytes, RS_BUF_FLAGS);
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void *)(out + (size_t)leaf0 * PIX * CIN), 0, nbytes, RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                xs[t][mt] = rs_load_x(xrs, rowoff, (tile0 + t) * 2048 + 64 * mt);
                *(f32x4 *)(img + pdst[t][mt]) = rs_relu(xs[t][mt]);
            }
        f32x4 acc[NT][2];
#define R32_BIAS(k) const f32x4 ba = *(const f32x4 *)(sbias + 32 * (k) + 4 * g), bb = *(const f32x4 *)(sbias + 32 * (k) + 16 + 4 * g)
        lds_barrier();
        r32_conv<NT, CIN>(img, frs, 0, CONV_BYTES, PW, abase, acc, wq);
        lds_barrier();
        {
            R32_BIAS(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) { *(f32x4 *)(img + pdst[t][0]) = rs_relu(acc[t][0] + ba); *(f32x4 *)(img + pdst[t][1]) = rs_relu(acc[t][1] + bb); }
        }
        lds_barrier();
        r32_conv<NT, CIN>(img, frs, CONV_BYTES, 2 * CONV_BYTES, PW, abase, acc, wq);
        lds_barrier();
        {
            R32_BIAS(1);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                xs[t][0] = (acc[t][0] + ba) + xs[t][0]; xs[t][1] = (acc[t][1] + bb) + xs[t][1];
                *(f32x4 *)(img + pdst[t][0]) = rs_relu(xs[t][0]); *(f32x4 *)(img + pdst[t][1]) = rs_relu(xs[t][1]);
            }
        }
        lds_barrier();
        r32_conv<NT, CIN>(img, frs, 2 * CONV_BYTES, 3 * CONV_BYTES, PW, abase, acc, wq);
        lds_barrier();
        {
            R32_BIAS(2);
#pragma unroll
            for (int t = 0; t < NT; ++t) { *(f32x4 *)(img + pdst[t][0]) = rs_relu(acc[t][0] + ba); *(f32x4 *)(img + pdst[t][1]) = rs_relu(acc[t][1] + bb); }
        }
        lds_barrier();
        r32_conv<NT, CIN>(img, frs, 3 * CONV_BYTES, 0, PW, abase, acc, wq);
        {
            R32_BIAS(3);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = (acc[t][0] + ba) + xs[t][0]; acc[t][1] = (acc[t][1] + bb) + xs[t][1];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][0]), ors, rowoff, (tile0 + t) * 2048, RS_STREAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[t][1]), ors, rowoff, (tile0 + t) * 2048 + 64, RS_STREAM_AUX);
            }
        }
#undef R32_BIAS
        if (out_relu != nullptr) {
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(out_relu + (size_t)leaf0 * PIX * CIN), 0, nbytes, RS_BUF_FLAGS);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t][0])), rrs, rowoff, (tile0 + t) * 2048, RS_STREAM_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs_relu(acc[t][1])), rrs, rowoff, (tile0 + t) * 2048 + 64, RS_STREAM_AUX);
            }
        }
        lds_barrier();
    }
}

// Stage entry (3x3 convolution CIN -> 32 + bias + max_pool2d(3, 2, 1)) with one workgroup per IMGW leaves: convolution as above, the
// output + bias goes to an LDS staging copy of rows of 36 floats (it overwrites the consumed input image), the whole workgroup pools.
template <int NT, int CIN, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k_convpool32_wg(const float *__restrict__ x, const float *__restrict__ frag, const float *__restrict__ bias,
                                                          float *__restrict__ out, long long B, int S_h, int S_w, int IMGW, int wg_floats,
                                                          const int *__restrict__ nrows_dev) {
    constexpr int XQ = CIN / 16;
    extern __shared__ __attribute__((aligned(16))) float rb_lds[];
    const int lane = lane_id(), wv = wave_in_block();
    if (nrows_dev) { const long long n = *nrows_dev; if (n < B) B = n; }
    if ((long long)blockIdx.x * IMGW >= B) return;
    constexpr int PS = r32_ps(CIN);
    const int PW = r32_pw(S_w), PIX = S_h * S_w, IMGP = r32_imgp(S_h, S_w), MP = IMGW * PIX;
    const int Hp = (S_h + 1) >> 1, Wp = (S_w + 1) >> 1, PP = Hp * Wp;
    float *img = rb_lds;  // wg_floats: the padded input images + dummy pixel, later the staging rows
    {
        float4 *z4 = (float4 *)img;
        for (int i = threadIdx.x; i < wg_floats / 4; i += 64 * WAVES) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int n = lane & 15, g = lane >> 4;
    int abase[NT], pin[NT][XQ], m_row[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int m = (wv * NT + t) * 16 + n, mm = m < MP ? m : 0, im = mm / PIX, pq = mm - im * PIX, r = pq / S_w, c = pq - r * S_w;
        const int pix = im * IMGP + (r + 1) * PW + c + 1, pc = m < MP ? pix : IMGW * IMGP;
        abase[t] = (pix - (PW + 1)) * PS + CIN / 4 * g;
        m_row[t] = m;
#pragma unroll
        for (int q = 0; q < XQ; ++q) pin[t][q] = pc * PS + 4 * (4 * q + g);
    }
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 9 * CIN * 32 * 4, RS_BUF_FLAGS);
    const f32x4 ba = *(const f32x4 *)(bias + 4 * g), bb = *(const f32x4 *)(bias + 16 + 4 * g);
    const int rowoff = n * (CIN * 4) + g * 16, tile0 = wv * NT;
    f32x4 wq[3][2];
    wq[0][0] = rs_load_b(frs, lane * 16, 0); wq[0][1] = rs_load_b(frs, lane * 16, 1024);
    wq[1][0] = rs_load_b(frs, lane * 16, 2048); wq[1][1] = rs_load_b(frs, lane * 16, 3072);
    const int cq = threadIdx.x & 7;
    lds_barrier();
    const long long stride_leaves = (long long)gridDim.x * IMGW;
    for (long long leaf0 = (long long)blockIdx.x * IMGW; leaf0 < B; leaf0 += stride_leaves) {
        const int nimg = (int)(B - leaf0 < IMGW ? B - leaf0 : IMGW);
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (size_t)leaf0 * PIX * CIN), 0, nimg * PIX * CIN * 4, RS_BUF_FLAGS);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < XQ; ++q) *(f32x4 *)(img + pin[t][q]) = rs_load_x(xrs, rowoff, (tile0 + t) * 16 * CIN * 4 + 64 * q);
        f32x4 acc[NT][2];
        lds_barrier();
        r32_conv<NT, CIN>(img, frs, 0, 0, PW, abase, acc, wq);
        lds_barrier();
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (m_row[t] < 16 * ((MP + 15) / 16)) {  // rows past the last (partly filled) tile would land beyond the staging area
                *(f32x4 *)(img + m_row[t] * 36 + 4 * g) = acc[t][0] + ba;
                *(f32x4 *)(img + m_row[t] * 36 + 16 + 4 * g) = acc[t][1] + bb;
            }
        lds_barrier();
        float4 *o4 = (float4 *)(out + (size_t)leaf0 * PP * 32);
        for (int pq = threadIdx.x >> 3; pq < nimg * PP; pq += 8 * WAVES) {
            const int im = pq / PP, pp = pq - im * PP, pr = pp / Wp, px = pp - pr * Wp;
            const int ctr = im * PIX + 2 * pr * S_w + 2 * px;
            const int ou = pr > 0 ? -S_w : 0, od = 2 * pr + 1 < S_h ? S_w : 0, ol = px > 0 ? -1 : 0, orr = 2 * px + 1 < S_w ? 1 : 0;
            const float4 *row = (const float4 *)(img + ctr * 36) + cq;
            const float4 v0 = row[0], v1 = row[ol * 9], v2 = row[orr * 9], v3 = row[ou * 9], v4 = row[(ou + ol) * 9], v5 = row[(ou + orr) * 9],
                         v6 = row[od * 9], v7 = row[(od + ol) * 9], v8 = row[(od + orr) * 9];
            float4 m;
            m.x = fmaxf(fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x)), fmaxf(fmaxf(v4.x, v5.x), fmaxf(fmaxf(v6.x, v7.x), v8.x)));
            m.y = fmaxf(fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y)), fmaxf(fmaxf(v4.y, v5.y), fmaxf(fmaxf(v6.y, v7.y), v8.y)));
            m.z = fmaxf(fmaxf(fmaxf(v0.z, v1.z), fmaxf(v2.z, v3.z)), fmaxf(fmaxf(v4.z, v5.z), fmaxf(fmaxf(v6.z, v7.z), v8.z)));
            m.w = fmaxf(fmaxf(fmaxf(v0.w, v1.w), fmaxf(v2.w, v3.w)), fmaxf(fmaxf(v4.w, v5.w), fmaxf(fmaxf(v6.w, v7.w), v8.w)));
            o4[pq * 8 + cq] = m;
        }
        lds_barrier();
        if (leaf0 + stride_leaves < B) {  // the staging copy overwrote the padded images: borders back to zero for the next task
            float4 *z4 = (float4 *)img;
            for (int i = threadIdx.x; i < wg_floats / 4; i += 64 * WAVES) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            lds_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct rp_ctx {
    rp_config cfg;
    DP d;
    hipStream_t stream;
    bool row64;
    std::string err;
    std::vector<void *> allocs;
    int64_t bytes;
    int64_t fin_popped;
    u8 *pool_wh;
    int *pool_area, *pool_max_h;
    int64_t pool_cap;
    int n_cu = 256;                   // compute units and LDS bytes per CU of the device (hipDeviceProp, read once in rp_create)
    size_t lds_per_cu = 160 * 1024;
    int compact_rows = 0;             // rp_set_compact_rows: rp_search_step(ctx, NULL) lists the waiting slots on the device too
    const int *nn_rows_dev = nullptr;  // row limit of the rp_nn_* stage kernels (eval_count) while compact rows are on
};

static std::string g_create_error;

static int fail(rp_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}
#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T> static int dev_alloc(rp_ctx *ctx, T **out, size_t n, bool zero = true) {
    void *ptr = nullptr;
    size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    HIPCHK(ctx, hipMalloc(&ptr, bytes));
    ctx->allocs.push_back(ptr);
    ctx->bytes += (int64_t)bytes;
    if (zero) HIPCHK(ctx, hipMemsetAsync(ptr, 0, bytes, ctx->stream));
    *out = (T *)ptr;
    return RP_OK;
}
#define ALLOC(ctx, ptr, n)                       \
    do {                                         \
        int rc_ = dev_alloc(ctx, &(ptr), (n));   \
        if (rc_ != RP_OK) return rc_;            \
    } while (0)

// kernels that can create nodes stage a node's legal moves in LDS: WAVES_PER_BLOCK runs of A actions
#define STAGE_BYTES(d) ((size_t)WAVES_PER_BLOCK * (size_t)(d).A * sizeof(u16))

// Leaves per wave for the persistent stage kernels.  Their grid has 2 048 waves (2 workgroups per CU); a launch takes
// ceil(tasks / waves) rounds of nt pixel tiles.  Among the group sizes that fit LDS, take the one with the fewest tile-rounds -- for
// all B rows and, weighted 3 : 1, for the ~92 % of them that hold a leaf in an average wave (3x3 images: 6 leaves = 4 tiles x 3
// rounds, 5 leaves = 3 tiles x 3 rounds: a tenth less time at 30 000 leaves).
static int pick_leaves_per_wave(long long B, int PIX, int imgw_max) {
    const long long waves = 2048, typical = std::max<long long>(1, (long long)(0.92 * (double)B));
    long long best = -1;
    int pick = std::max(1, imgw_max);
    for (int k = std::max(1, imgw_max); k >= 1; --k) {
        const long long ntk = (k * PIX + 15) / 16;
        auto rounds = [&](long long rows) { const long long tasks = (rows + k - 1) / k; return (tasks + waves - 1) / waves; };
        const long long cost = (rounds(B) + 3 * rounds(typical)) * ntk;
        if (best < 0 || cost < best) { best = cost; pick = k; }
    }
    return pick;
}

// Leaves per WORKGROUP for the _wg stage kernels.  A CU works through its workgroups' pixel tiles at a fixed rate, so a launch takes
// (workgroups on the busiest CU) x (tile slots of a workgroup = waves x tiles per wave): few large groups fill their tiles best but
// quantise badly over the CUs (1 024 leaves in groups of 3 = 342 workgroups = 2 rounds on 86 CUs, 1 on the rest).  Cost for all B rows
// plus, weighted 3 : 1, the ~92 % of them that hold a leaf in an average wave -- as pick_leaves_per_wave does for the wave kernels.
static long long wg_group_cost(long long B, int k, int tile_slots, int n_cu) {
    const long long typical = std::max<long long>(1, (long long)(0.92 * (double)B));
    auto rounds = [&](long long rows) { const long long tasks = (rows + k - 1) / k; return (tasks + n_cu - 1) / n_cu; };
    return (rounds(B) + 3 * rounds(typical)) * tile_slots;
}

static int grid_for(long long waves) { return (int)((waves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK); }

// Dynamic LDS above the default 64 KB limit has to be allowed per kernel; a size the device cannot give is an argument error with
// the figures in the message, not a launch failure later.
static int allow_lds(rp_ctx *ctx, const void *fn, size_t lds, const char *what) {
    if (lds > ctx->lds_per_cu) return fail(ctx, RP_ERR_ARG, "%s: needs %zu bytes of LDS per workgroup, the device has %zu per CU", what, lds, ctx->lds_per_cu);
    if (lds <= 64 * 1024) return RP_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s", what, lds, hipGetErrorString(e));
    return RP_OK;
}

static int check_device_error(rp_ctx *ctx) {
    int e = 0;
    HIPCHK(ctx, hipMemcpyAsync(&e, ctx->d.error, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (e == 0) return RP_OK;
    (void)hipMemsetAsync(ctx->d.error, 0, sizeof(int), ctx->stream);
    static const char *names[] = {"", "node arena overflow (raise node_cap)", "legal-move arena overflow (raise edge_cap)",
                                  "transposition table full", "action is not a legal move of the (expanded) root", "search path broken",
                                  "finished-episode ring overflow", "replay buffer full (raise max_examples / max_sparse)",
                                  "visited-edge arena overflow (raise vis_cap)", "replay example index or sparse entry outside the packed arrays"};
    int code = (e == ERR_BAD_ACTION) ? RP_ERR_ASSERT : (e == ERR_PATH ? RP_ERR_STATE : (e == ERR_BAD_EXAMPLE ? RP_ERR_ARG : RP_ERR_CAPACITY));
    return fail(ctx, code, "device error %d: %s", e, e < 10 ? names[e] : "?");
}

// leaves and combine schedule of NumPy's pairwise sum over n elements (oracle: pairwise_sum)
static int build_plan(int lo, int n, std::vector<int> &llo, std::vector<int> &ln, std::vector<int> &sd, std::vector<int> &ss) {
    if (n <= 128) {
        llo.push_back(lo); ln.push_back(n);
        return (int)llo.size() - 1;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int a = build_plan(lo, n2, llo, ln, sd, ss);
    int b = build_plan(lo + n2, n - n2, llo, ln, sd, ss);
    sd.push_back(a); ss.push_back(b);
    return a;
}

extern "C" int rp_version(void) { return RP_ABI_VERSION; }

extern "C" const char *rp_last_error(const rp_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int64_t rp_device_bytes(const rp_ctx *ctx) { return ctx ? ctx->bytes : 0; }

extern "C" void rp_destroy(rp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocs) (void)hipFree(p);
    delete ctx;
}

extern "C" int rp_create(const rp_config *cfg, rp_ctx **out) {
    if (!cfg || !out) return fail(nullptr, RP_ERR_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != RP_ABI_VERSION) return fail(nullptr, RP_ERR_ARG, "abi_version %d != %d", cfg->abi_version, RP_ABI_VERSION);
    if (cfg->W < 1 || cfg->W > 64 || cfg->H < 1 || cfg->H > 64 || cfg->N < 1 || cfg->N > 128)
        return fail(nullptr, RP_ERR_ARG, "limits: 1<=W<=64, 1<=H<=64, 1<=N<=128 (got %d %d %d)", cfg->W, cfg->H, cfg->N);
    if (cfg->games < 1 || cfg->sims < 0) return fail(nullptr, RP_ERR_ARG, "games >= 1 and sims >= 0 required");
    if (!(cfg->cpuct > 0.0)) return fail(nullptr, RP_ERR_ARG, "cpuct must be positive (got %g): the best unvisited move of a node is kept as the one with the largest prior, which is the PUCT winner among the unvisited moves only for cpuct > 0", cfg->cpuct);
    if ((int64_t)cfg->sims * (cfg->N + 1) >= (int64_t)NSA_MASK) return fail(nullptr, RP_ERR_ARG, "sims too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device)
        return fail(nullptr, RP_ERR_DEVICE, "no HIP device %d (found %d): the engine has no CPU fallback", cfg->device, ndev);
    rp_ctx *ctx = new rp_ctx();
    ctx->cfg = *cfg;
    ctx->bytes = 0;
    ctx->fin_popped = 0;
    ctx->pool_wh = nullptr; ctx->pool_area = nullptr; ctx->pool_max_h = nullptr; ctx->pool_cap = 0;
    ctx->stream = (hipStream_t)cfg->stream;
    hipError_t e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, RP_ERR_DEVICE, "hipSetDevice: %s", hipGetErrorString(e)); }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
            if (prop.multiProcessorCount > 0) ctx->n_cu = prop.multiProcessorCount;
            // gfx950 has 160 KB of LDS per CU whatever the runtime's property says (earlier CDNA parts: 64 KB, which is also what
            // some runtimes report for every part); the larger of the two is the limit the launches are checked against
            size_t lds = strstr(prop.gcnArchName, "gfx950") ? (size_t)160 * 1024 : (size_t)64 * 1024;
            if (prop.maxSharedMemoryPerMultiProcessor > lds) lds = prop.maxSharedMemoryPerMultiProcessor;
            ctx->lds_per_cu = lds;
        }
    }
    DP &d = ctx->d;
    memset(&d, 0, sizeof d);
    d.W = cfg->W; d.H = cfg->H; d.N = cfg->N; d.A = cfg->W * cfg->N; d.G = cfg->games; d.sims = cfg->sims;
    d.cpuct = cfg->cpuct; d.seed = cfg->seed; d.tie_salt = cfg->tie_salt; d.move_rule = cfg->move_rule;
    d.node_cap = cfg->node_cap > 0 ? cfg->node_cap : cfg->sims * (cfg->N + 1) + 2;
    // legal moves per node: measured means are 8 of 80 actions (10x10/8), 54 of 640 (20x20/32) and 460 of 6 400 (50x50/128) -- about A / 12;
    // the automatic size (nothing is recycled by default: the API classes may re-root anywhere) allows A / 8, at least min(A, 96)
    d.edge_cap = cfg->edge_cap > 0 ? cfg->edge_cap : (int)std::min<int64_t>((int64_t)d.node_cap * std::max(std::min(d.A, 96), d.A / 8) + d.A, (int64_t)0x7FFF0000);
    d.vis_cap = cfg->vis_cap > 0 ? cfg->vis_cap : 6 * d.node_cap + 64;
    // level arenas: a chunk must hold the largest run (A legal moves / a visited block of up to A entries)
    auto pow2_at_least = [](int x) { int c = 1; while (c < x) c *= 2; return c; };
    d.pchunk = std::max(4096, pow2_at_least(d.A));
    d.vchunk = std::max(1024, pow2_at_least(d.A));
    // automatic sizes: every level keeps one partly filled chunk open, so add a chunk per level to the packed estimate
    d.n_pchunks = (d.edge_cap + d.pchunk - 1) / d.pchunk + (cfg->edge_cap > 0 ? 0 : cfg->N + 1);
    d.n_vchunks = (d.vis_cap + d.vchunk - 1) / d.vchunk + (cfg->vis_cap > 0 ? 0 : cfg->N + 1);
    d.n_pchunks = std::max(d.n_pchunks, 2); d.n_vchunks = std::max(d.n_vchunks, 2);
    if (d.n_pchunks > 0xFFFE || d.n_vchunks > 0xFFFE) { delete ctx; return fail(nullptr, RP_ERR_ARG, "arena too large for 16-bit chunk ids"); }
    d.edge_cap = d.n_pchunks * d.pchunk;
    d.vis_cap = d.n_vchunks * d.vchunk;
    d.reclaim = cfg->reclaim ? 1 : 0;
    d.rows_identity = 0;
    int tc = 64;
    while (tc < 2 * d.node_cap) tc *= 2;
    d.table_cap = tc;
    ctx->row64 = cfg->W > 32;
    d.RW = ctx->row64 ? 2 : 1;
    d.RMW = (cfg->N + 31) / 32;
    d.KW = cfg->H * d.RW + d.RMW;
    if (ctx->row64 && (d.KW & 1)) d.KW++;  // keep 64-bit rows 8-byte aligned
    d.magicW = (u32)(((1u << 20) + cfg->W - 1) / cfg->W);
    d.Hp = (cfg->H + 1) / 2; d.Wp = (cfg->W + 1) / 2;
    for (u32 a = 0; a < 8192u; ++a)
        if (((a * d.magicW) >> 20) != a / (u32)cfg->W) { delete ctx; return fail(nullptr, RP_ERR_ARG, "internal: division magic"); }
    const size_t G = (size_t)d.G, N = (size_t)d.N;
    int rc = RP_OK;
    auto A_ = [&](auto &ptr, size_t n) { if (rc == RP_OK) rc = dev_alloc(ctx, &ptr, n); };
    A_(d.item_wh, G * N * 2); A_(d.total_area, G); A_(d.max_h, G); A_(d.bl, G); A_(d.has_buf, G);
    A_(d.root, G); A_(d.n_nodes, G); A_(d.phase, G); A_(d.sims_done, G); A_(d.moves, G); A_(d.episode, G);
    A_(d.leaf_node, G); A_(d.path_len, G); A_(d.path_edge, G * N); A_(d.path_node, G * N); A_(d.game_row, G);
    A_(d.last_outcome, G); A_(d.last_score, G); A_(d.last_v, G); A_(d.last_vkind, G);
    A_(d.eval_count, 1); A_(d.eval_slot, G);
    {   // the slot slabs: one allocation, regions 256-byte aligned, hottest first (headers, hash table, visited blocks, keys, legal-move runs)
        size_t off = 0;
        auto region = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t o_hdr = region((size_t)d.node_cap * sizeof(NodeHdr)), o_table = region((size_t)d.table_cap * 8);
        const size_t o_vis = region((size_t)d.vis_cap * sizeof(VisEntry));
        const size_t o_key = region((size_t)d.node_cap * d.KW * 4), o_pAct = region((size_t)d.edge_cap * 2), o_pPi = region((size_t)d.edge_cap * 4);
        d.slab_stride = off;
        u8 *slab = nullptr;
        A_(slab, G * d.slab_stride);
        d.hdr = (NodeHdr *)(slab + o_hdr); d.table = (u64 *)(slab + o_table);
        d.vis = (VisEntry *)(slab + o_vis);
        d.key = (u32 *)(slab + o_key); d.pAct = (u16 *)(slab + o_pAct); d.pPi = (float *)(slab + o_pPi);
    }
    A_(d.pa_cur, G * (N + 1)); A_(d.pa_head, G * (N + 1)); A_(d.pa_used, G * (N + 1)); A_(d.pa_next, G * d.n_pchunks); A_(d.pa_stack, G * d.n_pchunks); A_(d.pa_tf, G * 2);
    A_(d.va_cur, G * (N + 1)); A_(d.va_head, G * (N + 1)); A_(d.va_used, G * (N + 1)); A_(d.va_next, G * d.n_vchunks); A_(d.va_stack, G * d.n_vchunks); A_(d.va_tf, G * 2);
    A_(d.peak_chunks, G * 2 + 2);
    A_(d.g_bl, 1); A_(d.g_has_buf, 1); A_(d.counters, CNT_N); A_(d.slot_cnt, G * CNT_N); A_(d.error, 1);
    d.fin_cap = (int)std::max<size_t>(4 * G, 1024);
    A_(d.fin_count, 1); A_(d.fin_episode, d.fin_cap); A_(d.fin_outcome, d.fin_cap); A_(d.fin_moves, d.fin_cap); A_(d.fin_score, d.fin_cap);
    d.auto_restart = cfg->auto_restart; d.max_examples = cfg->max_examples > 0 ? cfg->max_examples : 0;
    A_(d.next_instance, 1); A_(d.ex_count, 1); A_(d.ex_sp_cursor, 1); A_(d.slot_ex, G * N);
    {
        PoolDesc *pd_dev = nullptr;
        A_(pd_dev, 1);
        if (rc == RP_OK && hipMemset(pd_dev, 0, sizeof(PoolDesc)) != hipSuccess) rc = RP_ERR_DEVICE;
        d.pool_desc = pd_dev;
    }
    if (d.max_examples > 0) {
        A_(d.ex_key, (size_t)d.max_examples * d.KW); A_(d.ex_wh, (size_t)d.max_examples * N * 2);
        // sparse visit counts: at most min(A, sims + 1) root edges are ever visited; the automatic pool allows 64 per example on average
        // (measured 20-50 at 20x20 / 32 / 400 sims) and overflow is reported as RP_ERR_CAPACITY like the example count itself
        d.sp_cap = cfg->max_sparse > 0 ? cfg->max_sparse : d.max_examples * (long long)std::min(std::min(d.A, std::max(cfg->sims, 1) + 1), 64);
        if (d.sp_cap > 0x7FFFFFFFLL) rc = fail(nullptr, RP_ERR_ARG, "replay buffer: %lld (action, count) pairs exceed 2^31 - 1", d.sp_cap);
        A_(d.ex_sp_off, (size_t)d.max_examples); A_(d.ex_sp_n, (size_t)d.max_examples); A_(d.ex_sp_act, (size_t)d.sp_cap); A_(d.ex_sp_cnt, (size_t)d.sp_cap);
        A_(d.ex_value, (size_t)d.max_examples);
        A_(d.ex_episode, (size_t)d.max_examples); A_(d.ex_move, (size_t)d.max_examples);
    }
    std::vector<int> llo, ln, sd, ss;
    build_plan(0, d.A, llo, ln, sd, ss);
    if ((int)llo.size() > MAX_LEAVES) rc = fail(nullptr, RP_ERR_ARG, "action space too large");
    size_t L = llo.size(), S = sd.size();
    if (rc != RP_OK) {
        std::string msg = ctx->err.empty() ? g_create_error : ctx->err;
        for (void *p : ctx->allocs) (void)hipFree(p);
        delete ctx;
        return fail(nullptr, rc, "rp_create: %s (needs about %.1f GiB of HBM)", msg.c_str(),
                    (double)(G * ((size_t)d.node_cap * (32 + 4 * d.KW) + (size_t)d.edge_cap * 6 + (size_t)d.vis_cap * sizeof(VisEntry) + (size_t)d.table_cap * 8)) / (1 << 30));
    }
    // no chunk is open before the first episode begins
    (void)hipMemsetAsync(d.pa_cur, 0xFF, G * (N + 1) * sizeof(u16), ctx->stream); (void)hipMemsetAsync(d.pa_head, 0xFF, G * (N + 1) * sizeof(u16), ctx->stream);
    (void)hipMemsetAsync(d.va_cur, 0xFF, G * (N + 1) * sizeof(u16), ctx->stream); (void)hipMemsetAsync(d.va_head, 0xFF, G * (N + 1) * sizeof(u16), ctx->stream);
    d.n_leaves = (int)L;
    for (size_t k = 0; k < L; ++k) { d.leaf_lo[k] = (u16)llo[k]; d.leaf_n[k] = (u8)ln[k]; }
    for (size_t k = 0; k < S; ++k) { d.sched_dst[k] = (u8)sd[k]; d.sched_src[k] = (u8)ss[k]; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        for (void *p : ctx->allocs) (void)hipFree(p);
        delete ctx;
        return fail(nullptr, RP_ERR_DEVICE, "rp_create: device initialisation failed");
    }
    if (STAGE_BYTES(d) + 9 * 1024 > 64 * 1024) {  // A > ~6 900: static (8 KB) + staging LDS pass the default 64 KB limit
        const size_t lim = STAGE_BYTES(d) + 9 * 1024;
        const void *fns[5];
        if (ctx->row64) { fns[0] = d.N > 64 ? (const void *)k_search<u64, true> : (const void *)k_search<u64, false>; fns[1] = (const void *)k_moves<u64>; fns[2] = (const void *)k_set_roots<u64>; fns[3] = (const void *)k_advance<u64>; fns[4] = (const void *)k_pool_begin<u64>; }
        else { fns[0] = d.N > 64 ? (const void *)k_search<u32, true> : (const void *)k_search<u32, false>; fns[1] = (const void *)k_moves<u32>; fns[2] = (const void *)k_set_roots<u32>; fns[3] = (const void *)k_advance<u32>; fns[4] = (const void *)k_pool_begin<u32>; }
        for (const void *fn : fns) {
            const int rc2 = allow_lds(ctx, fn, lim, "rp_create (legal-move staging of the tree kernels)");
            if (rc2 != RP_OK) {
                const std::string msg = ctx->err;
                for (void *p : ctx->allocs) (void)hipFree(p);
                delete ctx;
                return fail(nullptr, rc2, "%s", msg.c_str());
            }
        }
    }
    *out = ctx;
    return RP_OK;
}

#define DISPATCH(ctx, kernel, grid, ...)                                                                   \
    do {                                                                                                   \
        if ((ctx)->row64) hipLaunchKernelGGL(kernel<u64>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, (ctx)->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel<u32>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, (ctx)->stream, __VA_ARGS__);              \
        hipError_t le_ = hipGetLastError();                                                                \
        if (le_ != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "launch of %s failed: %s", #kernel, hipGetErrorString(le_)); \
    } while (0)

#define DISPATCH_STAGED(ctx, kernel, grid, ...)                                                            \
    do {                                                                                                   \
        if ((ctx)->row64) hipLaunchKernelGGL(kernel<u64>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), STAGE_BYTES((ctx)->d), (ctx)->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel<u32>, dim3(grid), dim3(64 * WAVES_PER_BLOCK), STAGE_BYTES((ctx)->d), (ctx)->stream, __VA_ARGS__);              \
        hipError_t le_ = hipGetLastError();                                                                \
        if (le_ != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "launch of %s failed: %s", #kernel, hipGetErrorString(le_)); \
    } while (0)

// scratch device copies of host inputs for the stateless calls
struct Scratch {
    rp_ctx *ctx;
    std::vector<void *> ptrs;
    explicit Scratch(rp_ctx *c) : ctx(c) {}
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T> T *up(const T *host, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        if (host && hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return nullptr;
        return (T *)p;
    }
};
#define NEED(ptr) do { if (!(ptr)) return fail(ctx, RP_ERR_DEVICE, "scratch allocation / upload failed"); } while (0)

extern "C" int rp_valid_moves(rp_ctx *ctx, int64_t B, const uint64_t *rows, const uint8_t *remaining, const uint8_t *item_wh,
                              uint8_t *mask_out, int32_t *n_valid_out) {
    if (!ctx || B < 0 || !rows || !remaining || !item_wh || !mask_out) return fail(ctx, RP_ERR_ARG, "rp_valid_moves: bad argument");
    if (B == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    u64 *drows = s.up((const u64 *)rows, (size_t)B * d.H); NEED(drows);
    u8 *drem = s.up(remaining, (size_t)B * d.N); NEED(drem);
    u8 *dwh = s.up(item_wh, (size_t)B * d.N * 2); NEED(dwh);
    u8 *dmask = s.up((const u8 *)nullptr, (size_t)B * d.A); NEED(dmask);
    int *dnv = s.up((const int *)nullptr, (size_t)B); NEED(dnv);
    DISPATCH(ctx, k_valid_moves, grid_for(B), d, (long long)B, drows, drem, dwh, dmask, dnv);
    HIPCHK(ctx, hipMemcpyAsync(mask_out, dmask, (size_t)B * d.A, hipMemcpyDeviceToHost, ctx->stream));
    if (n_valid_out) HIPCHK(ctx, hipMemcpyAsync(n_valid_out, dnv, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_apply_move(rp_ctx *ctx, int64_t B, const uint64_t *rows, const uint8_t *remaining, const uint8_t *item_wh,
                             const int32_t *action, uint64_t *rows_out, uint8_t *remaining_out, int32_t *status_out) {
    if (!ctx || B < 0 || !rows || !remaining || !item_wh || !action || !rows_out || !remaining_out || !status_out)
        return fail(ctx, RP_ERR_ARG, "rp_apply_move: bad argument");
    if (B == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    u64 *drows = s.up((const u64 *)rows, (size_t)B * d.H); NEED(drows);
    u8 *drem = s.up(remaining, (size_t)B * d.N); NEED(drem);
    u8 *dwh = s.up(item_wh, (size_t)B * d.N * 2); NEED(dwh);
    int *dact = s.up(action, (size_t)B); NEED(dact);
    u64 *drows_o = s.up((const u64 *)nullptr, (size_t)B * d.H); NEED(drows_o);
    u8 *drem_o = s.up((const u8 *)nullptr, (size_t)B * d.N); NEED(drem_o);
    int *dst = s.up((const int *)nullptr, (size_t)B); NEED(dst);
    DISPATCH(ctx, k_apply_move, grid_for(B), d, (long long)B, drows, drem, dwh, dact, drows_o, drem_o, dst);
    HIPCHK(ctx, hipMemcpyAsync(rows_out, drows_o, (size_t)B * d.H * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(remaining_out, drem_o, (size_t)B * d.N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(status_out, dst, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

// sorted[int(floor(len*alpha)) - 1] with Python's negative-index wrap (BinPackingGame.py:205-206)
static bool rank_threshold(const double *rewards, int n, double alpha, double *bl) {
    if (n <= 0) return false;
    std::vector<double> s(rewards, rewards + n);
    std::sort(s.begin(), s.end());
    int idx = (int)floor((double)n * alpha) - 1;
    if (idx < 0) idx += n;
    if (idx >= n) idx = n - 1;
    *bl = s[idx];
    return true;
}

extern "C" int rp_game_ended(rp_ctx *ctx, int64_t B, const uint64_t *rows, const uint8_t *remaining, const uint8_t *item_wh,
                             const int32_t *total_area, const int32_t *max_h, const double *rewards, int32_t n_rewards, double alpha,
                             int32_t *ended_out, double *reward_out) {
    if (!ctx || B < 0 || !rows || !remaining || !item_wh || !total_area || !max_h || !ended_out || !reward_out || n_rewards < 0 ||
        (n_rewards > 0 && !rewards))
        return fail(ctx, RP_ERR_ARG, "rp_game_ended: bad argument");
    if (B == 0) return RP_OK;
    const DP &d = ctx->d;
    double bl = 0.0;
    bool has = rank_threshold(rewards, n_rewards, alpha, &bl);
    Scratch s(ctx);
    u64 *drows = s.up((const u64 *)rows, (size_t)B * d.H); NEED(drows);
    u8 *drem = s.up(remaining, (size_t)B * d.N); NEED(drem);
    u8 *dwh = s.up(item_wh, (size_t)B * d.N * 2); NEED(dwh);
    int *darea = s.up(total_area, (size_t)B); NEED(darea);
    int *dmh = s.up(max_h, (size_t)B); NEED(dmh);
    int *dend = s.up((const int *)nullptr, (size_t)B); NEED(dend);
    double *drew = s.up((const double *)nullptr, (size_t)B); NEED(drew);
    DISPATCH(ctx, k_game_ended, grid_for(B), d, (long long)B, drows, drem, dwh, darea, dmh, has ? 1 : 0, bl, dend, drew);
    HIPCHK(ctx, hipMemcpyAsync(ended_out, dend, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(reward_out, drew, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_set_rank_buffer(rp_ctx *ctx, const double *rewards, int32_t n) {
    if (!ctx || n < 0 || (n > 0 && !rewards)) return fail(ctx, RP_ERR_ARG, "rp_set_rank_buffer: bad argument");
    double bl = 0.0;
    int has = rank_threshold(rewards, n, ctx->cfg.alpha, &bl) ? 1 : 0;
    hipLaunchKernelGGL(k_fill_rank, dim3((ctx->d.G + 255) / 256), dim3(256), 0, ctx->stream, ctx->d, bl, has);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_begin_episodes(rp_ctx *ctx, int32_t first, int32_t count, const uint8_t *item_wh, const int32_t *total_area,
                                 const uint64_t *episode_id) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G || !item_wh || !total_area)
        return fail(ctx, RP_ERR_ARG, "rp_begin_episodes: bad argument");
    if (count == 0) return RP_OK;
    const DP &d = ctx->d;
    std::vector<int> mh(count, 0);
    std::vector<u64> ids(count);
    for (int k = 0; k < count; ++k) {
        for (int i = 0; i < d.N; ++i) {
            int w = item_wh[((size_t)k * d.N + i) * 2], h = item_wh[((size_t)k * d.N + i) * 2 + 1];
            if (w < 1 || w > d.W || h < 1 || h > d.H) return fail(ctx, RP_ERR_ARG, "item %d of episode %d has size %dx%d outside the %dx%d grid", i, k, w, h, d.W, d.H);
            mh[k] = std::max(mh[k], h);  // BinPackingGame.getInitItems: max_h over ALL items (BinPackingGame.py:41-50)
        }
        ids[k] = episode_id ? episode_id[k] : (u64)(first + k);
    }
    HIPCHK(ctx, hipMemcpyAsync(d.item_wh + (size_t)first * d.N * 2, item_wh, (size_t)count * d.N * 2, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d.total_area + first, total_area, (size_t)count * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d.max_h + first, mh.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d.episode + first, ids.data(), (size_t)count * sizeof(u64), hipMemcpyHostToDevice, ctx->stream));
    DISPATCH_STAGED(ctx, k_set_roots, grid_for(count), d, (int)first, (int)count, (const u64 *)nullptr, (const u8 *)nullptr, 1);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host vectors go out of scope
    return check_device_error(ctx);
}

extern "C" int rp_set_roots(rp_ctx *ctx, int32_t first, int32_t count, const uint64_t *rows, const uint8_t *remaining) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G || !rows || !remaining) return fail(ctx, RP_ERR_ARG, "rp_set_roots: bad argument");
    if (count == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    u64 *drows = s.up((const u64 *)rows, (size_t)count * d.H); NEED(drows);
    u8 *drem = s.up(remaining, (size_t)count * d.N); NEED(drem);
    DISPATCH_STAGED(ctx, k_set_roots, grid_for(count), d, (int)first, (int)count, (const u64 *)drows, (const u8 *)drem, 0);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

extern "C" int rp_set_stream(rp_ctx *ctx, void *stream) {
    if (!ctx) return RP_ERR_ARG;
    ctx->stream = (hipStream_t)stream;
    ctx->cfg.stream = stream;
    return RP_OK;
}

extern "C" int rp_set_step_cap(rp_ctx *ctx, int32_t max_sims_per_step) {
    if (!ctx || max_sims_per_step < 0) return fail(ctx, RP_ERR_ARG, "rp_set_step_cap: bad argument");
    ctx->d.step_cap = max_sims_per_step;
    return RP_OK;
}

extern "C" int rp_set_compact_rows(rp_ctx *ctx, int32_t enable) {
    if (!ctx) return RP_ERR_ARG;
    ctx->compact_rows = enable ? 1 : 0;
    ctx->nn_rows_dev = enable ? ctx->d.eval_count : nullptr;
    return RP_OK;
}

extern "C" int rp_set_move_rule(rp_ctx *ctx, int32_t move_rule, int32_t onehot_examples) {
    if (!ctx || move_rule < RP_MOVE_EXTERNAL || move_rule > RP_MOVE_SAMPLE) return fail(ctx, RP_ERR_ARG, "rp_set_move_rule: bad argument");
    ctx->d.move_rule = move_rule;
    ctx->d.onehot_examples = onehot_examples ? 1 : 0;
    ctx->cfg.move_rule = move_rule;
    return RP_OK;
}

extern "C" int rp_set_sims(rp_ctx *ctx, int32_t sims) {
    if (!ctx || sims < 0 || (int64_t)sims * (ctx->d.N + 1) >= (int64_t)NSA_MASK) return fail(ctx, RP_ERR_ARG, "rp_set_sims: bad argument");
    ctx->d.sims = sims;  // the device view is passed by value at every launch
    ctx->cfg.sims = sims;
    return RP_OK;
}

extern "C" int rp_last_values(rp_ctx *ctx, int32_t first, int32_t count, double *v_out, int32_t *kind_out) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G || !v_out) return fail(ctx, RP_ERR_ARG, "rp_last_values: bad argument");
    HIPCHK(ctx, hipMemcpyAsync(v_out, ctx->d.last_v + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (kind_out) HIPCHK(ctx, hipMemcpyAsync(kind_out, ctx->d.last_vkind + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_search_step(rp_ctx *ctx, int32_t *n_leaves_out) {
    if (!ctx) return RP_ERR_ARG;
    DP &d = ctx->d;
    if (d.move_rule != RP_MOVE_EXTERNAL) DISPATCH_STAGED(ctx, k_moves, grid_for(d.G), d);
    {
        const dim3 grid(grid_for(d.G)), block(64 * WAVES_PER_BLOCK);
        const size_t lds = STAGE_BYTES(d);
        if (ctx->row64) { if (d.N > 64) hipLaunchKernelGGL((k_search<u64, true>), grid, block, lds, ctx->stream, d); else hipLaunchKernelGGL((k_search<u64, false>), grid, block, lds, ctx->stream, d); }
        else { if (d.N > 64) hipLaunchKernelGGL((k_search<u32, true>), grid, block, lds, ctx->stream, d); else hipLaunchKernelGGL((k_search<u32, false>), grid, block, lds, ctx->stream, d); }
        hipError_t le_ = hipGetLastError();
        if (le_ != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "launch of k_search failed: %s", hipGetErrorString(le_));
    }
    // Without a count request nothing is synchronised and evaluator row b belongs to slot b (fixed shapes for graph capture);
    // with one, the waiting slots are listed in slot order and rows follow that list.
    d.rows_identity = (n_leaves_out || ctx->compact_rows) ? 0 : 1;
    if (!d.rows_identity) {
        hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, ctx->stream, d);
        HIPCHK(ctx, hipGetLastError());
    }
    if (n_leaves_out) {
        int n = 0;
        HIPCHK(ctx, hipMemcpyAsync(&n, d.eval_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        int rc = check_device_error(ctx);  // synchronises
        if (rc != RP_OK) return rc;
        *n_leaves_out = n;
    }
    return RP_OK;
}

extern "C" int rp_leaf_planes(rp_ctx *ctx, float *planes_dev, int64_t capacity_rows) {
    if (!ctx || !planes_dev || capacity_rows < 0) return fail(ctx, RP_ERR_ARG, "rp_leaf_planes: bad argument");
    const DP &d = ctx->d;
    long long rows = std::min<long long>(capacity_rows, d.G);
    if (rows == 0) return RP_OK;
    DISPATCH(ctx, k_leaf_planes, grid_for(rows), d, planes_dev, (long long)capacity_rows);
    return RP_OK;
}

extern "C" int rp_leaf_count_async(rp_ctx *ctx, int32_t *count_host) {
    if (!ctx || !count_host) return fail(ctx, RP_ERR_ARG, "rp_leaf_count_async: bad argument");
    HIPCHK(ctx, hipMemcpyAsync(count_host, ctx->d.eval_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return RP_OK;
}

extern "C" int rp_stem_set_weights(rp_ctx *ctx, const float *conv_w_dev, const float *bias_dev) {
    if (!ctx || !conv_w_dev || !bias_dev) return fail(ctx, RP_ERR_ARG, "rp_stem_set_weights: bad argument");
    DP &d = ctx->d;
    if (!d.stemT) {
        ALLOC(ctx, d.stemT, (size_t)(d.N * 25 + 1) * STEM_C);  // + one all-zero row (ALLOC clears): the target of outputs an item does not reach
        ALLOC(ctx, d.stemTB, (size_t)512 * STEM_C);
        ALLOC(ctx, d.stemBias, (size_t)STEM_C);
        ALLOC(ctx, d.stemScale, (size_t)STEM_C);
    }
    const size_t tbl_lds = ((size_t)d.N * 25 + 512) * sizeof(double);  // 29.7 KB at N = 128
    hipLaunchKernelGGL(k_stem_tables, dim3(STEM_C), dim3(256), tbl_lds, ctx->stream, d.N, conv_w_dev, bias_dev, d.stemT, d.stemTB, d.stemBias, d.stemScale);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_leaf_stem(rp_ctx *ctx, float *out_dev, float *out_relu_dev, int64_t capacity_rows, int32_t channels_last) {
    if (!ctx || !out_dev || capacity_rows < 0) return fail(ctx, RP_ERR_ARG, "rp_leaf_stem: bad argument");
    const DP &d = ctx->d;
    if (!d.stemT) return fail(ctx, RP_ERR_STATE, "rp_leaf_stem: call rp_stem_set_weights first");
    long long rows = std::min<long long>(capacity_rows, d.G);
    if (rows == 0) return RP_OK;
    const size_t t_bytes = (size_t)(d.N * 25 + 1) * STEM_C * sizeof(int);  // the item table and its zero row
    const bool t_lds = t_bytes <= 64 * 1024;
    // LDS form: persistent workgroups, 3 per CU (3 x 51 KB of LDS at N = 32), a wave per leaf; L2 form: a workgroup per leaf
    const int grid = (int)std::min<long long>(t_lds ? grid_for(rows) : rows, t_lds ? ctx->n_cu * STEM_WAVES : 1 << 20);
    if (ctx->row64) {
        if (t_lds) hipLaunchKernelGGL((k_leaf_stem<u64, true>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), t_bytes, ctx->stream, d, out_dev, out_relu_dev, (long long)capacity_rows, (int)(channels_last != 0));
        else hipLaunchKernelGGL((k_leaf_stem<u64, false>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, ctx->stream, d, out_dev, out_relu_dev, (long long)capacity_rows, (int)(channels_last != 0));
    } else {
        if (t_lds) hipLaunchKernelGGL((k_leaf_stem<u32, true>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), t_bytes, ctx->stream, d, out_dev, out_relu_dev, (long long)capacity_rows, (int)(channels_last != 0));
        else hipLaunchKernelGGL((k_leaf_stem<u32, false>), dim3(grid), dim3(64 * WAVES_PER_BLOCK), 0, ctx->stream, d, out_dev, out_relu_dev, (long long)capacity_rows, (int)(channels_last != 0));
    }
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_bias_relu(rp_ctx *ctx, float *x_dev, const float *bias_dev, int64_t B, int32_t C, int32_t HW) {
    if (!ctx || !x_dev || !bias_dev || B < 0 || C < 1 || HW < 1) return fail(ctx, RP_ERR_ARG, "rp_nn_bias_relu: bad argument");
    long long n = (long long)B * C * HW;
    if (n % 4) return fail(ctx, RP_ERR_ARG, "rp_nn_bias_relu: element count must be a multiple of 4");
    if (n == 0) return RP_OK;
    hipLaunchKernelGGL(k_nn_bias_relu, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, x_dev, bias_dev, n / 4, (int)C, (int)HW);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_bias_residual(rp_ctx *ctx, const float *x_dev, const float *bias_dev, const float *res_dev, float *out_dev, float *out_relu_dev,
                                   int64_t B, int32_t C, int32_t HW) {
    if (!ctx || !x_dev || !bias_dev || !res_dev || !out_dev || B < 0 || C < 1 || HW < 1) return fail(ctx, RP_ERR_ARG, "rp_nn_bias_residual: bad argument");
    long long n = (long long)B * C * HW;
    if (n % 4) return fail(ctx, RP_ERR_ARG, "rp_nn_bias_residual: element count must be a multiple of 4");
    if (n == 0) return RP_OK;
    hipLaunchKernelGGL(k_nn_bias_residual, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, x_dev, bias_dev, res_dev, out_dev, out_relu_dev,
                       n / 4, (int)C, (int)HW);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_bias_pool(rp_ctx *ctx, const float *x_dev, const float *bias_dev, float *out_dev, float *out_relu_dev, int64_t B, int32_t C, int32_t H,
                               int32_t W, int32_t channels_last) {
    if (!ctx || !x_dev || !bias_dev || !out_dev || B < 0 || C < 1 || H < 1 || W < 1) return fail(ctx, RP_ERR_ARG, "rp_nn_bias_pool: bad argument");
    int Hp = (H + 1) / 2, Wp = (W + 1) / 2;
    long long n = (long long)B * C * Hp * Wp;
    if (n == 0) return RP_OK;
    if (channels_last && C % 4 == 0 && n / 4 < 0xFFFFFF00LL)
        hipLaunchKernelGGL(k_nn_bias_pool_nhwc4, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, ctx->stream, (const float4 *)x_dev, (const float4 *)bias_dev,
                           (float4 *)out_dev, (float4 *)out_relu_dev, (unsigned)(n / 4), (unsigned)(C / 4), (int)H, (int)W, Hp, Wp);
    else if (channels_last)
        hipLaunchKernelGGL(k_nn_bias_pool_nhwc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x_dev, bias_dev, out_dev, out_relu_dev, n, (int)C,
                           (int)H, (int)W, Hp, Wp);
    else
        hipLaunchKernelGGL(k_nn_bias_pool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, x_dev, bias_dev, out_dev, out_relu_dev, n, (int)C, (int)H,
                           (int)W, Hp, Wp);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_value_head(rp_ctx *ctx, const float *z_dev, const float *w_dev, const float *bias_dev, float *out_dev, int64_t B, int32_t K) {
    if (!ctx || !z_dev || !w_dev || !bias_dev || !out_dev || B < 0 || K < 4 || K % 4) return fail(ctx, RP_ERR_ARG, "rp_nn_value_head: bad argument (K a multiple of 4)");
    if (B == 0) return RP_OK;
    hipLaunchKernelGGL(k_nn_value_head, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, ctx->stream, (const float4 *)z_dev, (const float4 *)w_dev, bias_dev, out_dev,
                       (long long)B, (int)(K / 4));
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_pack_conv16(rp_ctx *ctx, const float *w_dev, float *frag_dev) {
    if (!ctx || !w_dev || !frag_dev) return fail(ctx, RP_ERR_ARG, "rp_nn_pack_conv16: bad argument");
    hipLaunchKernelGGL(k_pack_conv16, dim3(9), dim3(256), 0, ctx->stream, w_dev, frag_dev);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_resblock16(rp_ctx *ctx, const float *x_dev, const float *frag0_dev, const float *bias0_dev, const float *frag1_dev, const float *bias1_dev,
                                float *out_dev, float *out_relu_dev, int64_t B, int32_t H, int32_t W) {
    if (!ctx || !x_dev || !frag0_dev || !bias0_dev || !frag1_dev || !bias1_dev || !out_dev || B < 0 || H < 1 || W < 1 || H > 62 || W > 62)
        return fail(ctx, RP_ERR_ARG, "rp_nn_resblock16: bad argument");
    if (B == 0) return RP_OK;
    const size_t lds = ((size_t)4 * 2 * (H + 2) * (W + 2) * RB_STRIDE + (((size_t)H * W + 3) & ~(size_t)3)) * sizeof(float);
    { const int rc = allow_lds(ctx, (const void *)k_resblock16, lds, "rp_nn_resblock16"); if (rc != RP_OK) return rc; }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, ctx->lds_per_cu / lds));
    const int grid = (int)std::min<long long>((B + 3) / 4, (long long)ctx->n_cu * per_cu);
    hipLaunchKernelGGL(k_resblock16, dim3(grid), dim3(256), lds, ctx->stream, x_dev, frag0_dev, bias0_dev, frag1_dev, bias1_dev, out_dev, out_relu_dev, (long long)B,
                       (int)H, (int)W);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_resstage16(rp_ctx *ctx, const float *x_dev, const float *frag4_dev, const float *bias4_dev, float *out_dev, float *out_relu_dev, int64_t B,
                                int32_t H, int32_t W) {
    if (!ctx || !x_dev || !frag4_dev || !bias4_dev || !out_dev || B < 0 || H < 1 || W < 1 || H * W > 640)
        return fail(ctx, RP_ERR_ARG, "rp_nn_resstage16: bad argument (images of at most 640 pixels)");
    if (B == 0) return RP_OK;
    const int PIX = H * W;
    if (PIX > 128) {  // one workgroup per image, the pixel tiles dealt to its four waves (k_resstage16_wg)
        const int tiles = (PIX + 15) / 16, waves = tiles > 20 ? 8 : 4, nt = (tiles + waves - 1) / waves;
        const size_t lds = ((size_t)(H + 2) * (W + 2) * RS_STRIDE + RS_STRIDE) * sizeof(float);
        if (lds > ctx->lds_per_cu) return fail(ctx, RP_ERR_ARG, "rp_nn_resstage16: a %dx%d image needs %zu bytes of LDS, the device has %zu per CU", H, W, lds, ctx->lds_per_cu);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, ctx->lds_per_cu / lds));
        const dim3 grid((unsigned)std::min<long long>(B, (long long)ctx->n_cu * per_cu)), block(64 * waves);
#define RSW_LAUNCH(NT_, WV_)                                                                                                                        \
    case NT_ * 16 + WV_: {                                                                                                                          \
        const int rc_ = allow_lds(ctx, (const void *)k_resstage16_wg<NT_, WV_>, lds, "rp_nn_resstage16"); if (rc_ != RP_OK) return rc_;            \
        hipLaunchKernelGGL((k_resstage16_wg<NT_, WV_>), grid, block, lds, ctx->stream, x_dev, frag4_dev, bias4_dev, out_dev, out_relu_dev, (long long)B, (int)H, (int)W, ctx->nn_rows_dev); \
    } break;
        switch (nt * 16 + waves) {
            RSW_LAUNCH(3, 4) RSW_LAUNCH(4, 4) RSW_LAUNCH(5, 4) RSW_LAUNCH(3, 8) RSW_LAUNCH(4, 8) RSW_LAUNCH(5, 8)
            default: return fail(ctx, RP_ERR_ARG, "rp_nn_resstage16: unsupported image size");
        }
#undef RSW_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return RP_OK;
    }
    const size_t img_bytes = (size_t)(H + 2) * (W + 2) * RS_STRIDE * sizeof(float);
    // leaves per wave: as many as fit 8 pixel tiles (accumulators + the kept skip operand in registers) and two workgroups per CU
    int imgw_max = std::max(1, (16 * 8) / PIX);
    while (imgw_max > 1 && 4 * (imgw_max * img_bytes + RS_STRIDE * sizeof(float)) + 1024 > 78 * 1024) --imgw_max;
    const int imgw = pick_leaves_per_wave(B, PIX, imgw_max);
    const int nt = (imgw * PIX + 15) / 16;
    const size_t lds = (size_t)16 * nt * sizeof(int) + 4 * (imgw * img_bytes + RS_STRIDE * sizeof(float));
    const long long tasks = (B + imgw - 1) / imgw;
    static const int stage_wgs = getenv("RP_STAGE16_WGS") ? atoi(getenv("RP_STAGE16_WGS")) : 2;  // resident workgroups per CU (registers: 2 waves per SIMD)
    if (lds > ctx->lds_per_cu) return fail(ctx, RP_ERR_ARG, "rp_nn_resstage16: a %dx%d image needs %zu bytes of LDS per workgroup, the device has %zu per CU", H, W, lds, ctx->lds_per_cu);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)stage_wgs, ctx->lds_per_cu / lds));
    const dim3 grid((unsigned)std::min<long long>((tasks + 3) / 4, (long long)ctx->n_cu * per_cu)), block(256);
#define RS_LAUNCH(NT_)                                                                                                                              \
    case NT_:                                                                                                                                       \
        { const int rc_ = allow_lds(ctx, (const void *)k_resstage16<NT_>, lds, "rp_nn_resstage16"); if (rc_ != RP_OK) return rc_; }                \
        hipLaunchKernelGGL((k_resstage16<NT_>), grid, block, lds, ctx->stream, x_dev, frag4_dev, bias4_dev, out_dev, out_relu_dev, (long long)B, (int)H, (int)W, imgw, ctx->nn_rows_dev); \
        break;
    static const int rs16_tail = getenv("RP_STAGE16_TAIL") ? atoi(getenv("RP_STAGE16_TAIL")) : 1;  // 0: the tail pixels as a whole tile (A/B)
    const int tail_px = (imgw * PIX) & 15;
    if (rs16_tail && nt == 7 && tail_px >= 1 && tail_px <= 4) {  // six tiles + a tail of up to four pixels (10x10: 96 + 4)
        { const int rc_ = allow_lds(ctx, (const void *)k_resstage16<6, true>, lds, "rp_nn_resstage16"); if (rc_ != RP_OK) return rc_; }
        hipLaunchKernelGGL((k_resstage16<6, true>), grid, block, lds, ctx->stream, x_dev, frag4_dev, bias4_dev, out_dev, out_relu_dev, (long long)B, (int)H, (int)W, imgw, ctx->nn_rows_dev);
        HIPCHK(ctx, hipGetLastError());
        return RP_OK;
    }
    switch (nt) {
        RS_LAUNCH(1) RS_LAUNCH(2) RS_LAUNCH(3) RS_LAUNCH(4) RS_LAUNCH(5) RS_LAUNCH(6) RS_LAUNCH(7) RS_LAUNCH(8)
        default: return fail(ctx, RP_ERR_ARG, "rp_nn_resstage16: unsupported image size");
    }
#undef RS_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_pack_conv32(rp_ctx *ctx, const float *w_dev, float *frag_dev, int32_t Cin) {
    if (!ctx || !w_dev || !frag_dev || (Cin != 16 && Cin != 32)) return fail(ctx, RP_ERR_ARG, "rp_nn_pack_conv32: bad argument (Cin 16 or 32)");
    const int n = 9 * Cin * 32;
    hipLaunchKernelGGL(k_pack_conv32, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, w_dev, frag_dev, (int)Cin);
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_convpool32(rp_ctx *ctx, const float *x_dev, const float *frag_dev, const float *bias_dev, float *out_dev, int64_t B, int32_t Cin, int32_t H,
                                int32_t W) {
    if (!ctx || !x_dev || !frag_dev || !bias_dev || !out_dev || B < 0 || H < 1 || W < 1 || (Cin != 16 && Cin != 32) || H * W > (Cin == 16 ? 640 : 512))
        return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: bad argument (Cin 16: <= 640 pixels, Cin 32: <= 512 pixels)");
    if (B == 0) return RP_OK;
    if (H * W > (Cin == 16 ? 112 : 80)) {  // IMGW leaves per workgroup (k_convpool32_wg)
        const int PIXw = H * W, max_nt = Cin == 16 ? 10 : 8;  // tile rows per workgroup in units of 64: 40 tiles (Cin 16) / 32 tiles (Cin 32)
        const size_t img_px = (size_t)r32_imgp(H, W);
        auto floats_of = [&](int k) { return (std::max<size_t>((k * img_px + 1) * r32_ps(Cin), (size_t)16 * ((k * PIXw + 15) / 16) * 36) + 3) & ~(size_t)3; };
        int imgw = 0, nt = 0, waves = 4;
        long long best = -1;
        for (int k = 1; k * PIXw <= 64 * max_nt && floats_of(k) * sizeof(float) <= ctx->lds_per_cu; ++k) {
            const int tk = (k * PIXw + 15) / 16, wk = tk > 16 ? 8 : 4, ntk = (tk + wk - 1) / wk;
            const long long cost = wg_group_cost(B, k, wk * ntk, ctx->n_cu);
            if (best < 0 || cost < best) { best = cost; imgw = k; nt = ntk; waves = wk; }
        }
        if (imgw == 0) return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: a %dx%d image does not fit LDS", H, W);
        const size_t wf = floats_of(imgw), lds = wf * sizeof(float);
        const long long tasks = (B + imgw - 1) / imgw;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, ctx->lds_per_cu / lds));
        const dim3 grid((unsigned)std::min<long long>(tasks, (long long)ctx->n_cu * per_cu)), block(64 * waves);
#define CPW_LAUNCH(NT_, CIN_, WV_)                                                                                                                   \
    case (NT_ * 64 + CIN_) * 16 + WV_: {                                                                                                             \
        const int rc_ = allow_lds(ctx, (const void *)k_convpool32_wg<NT_, CIN_, WV_>, lds, "rp_nn_convpool32"); if (rc_ != RP_OK) return rc_;        \
        hipLaunchKernelGGL((k_convpool32_wg<NT_, CIN_, WV_>), grid, block, lds, ctx->stream, x_dev, frag_dev, bias_dev, out_dev, (long long)B,       \
                           (int)H, (int)W, imgw, (int)wf, ctx->nn_rows_dev);                                                                         \
    } break;
        switch ((nt * 64 + (int)Cin) * 16 + waves) {
            CPW_LAUNCH(2, 16, 4) CPW_LAUNCH(3, 16, 4) CPW_LAUNCH(4, 16, 4) CPW_LAUNCH(3, 16, 8) CPW_LAUNCH(4, 16, 8) CPW_LAUNCH(5, 16, 8)
            CPW_LAUNCH(2, 32, 4) CPW_LAUNCH(3, 32, 4) CPW_LAUNCH(4, 32, 4) CPW_LAUNCH(3, 32, 8) CPW_LAUNCH(4, 32, 8)
            default: return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: unsupported image size");
        }
#undef CPW_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return RP_OK;
    }
    const int PIX = H * W, max_tiles = Cin == 16 ? 7 : 5;
    const size_t img_pixels = (size_t)r32_imgp(H, W);
    int imgw = (16 * max_tiles) / PIX;
    const int PPn = ((H + 1) / 2) * ((W + 1) / 2);
    // a wave's region holds its padded images at 36 floats per pixel (x, then the convolution output in place) + the dummy pixel
    auto wave_floats = [&](int k) { return (size_t)(k * img_pixels + 1) * CP_PS; };
    auto lds_bytes = [&](int k) { return ((size_t)16 * ((k * PIX + 15) / 16) + (((size_t)k * PPn + 3) & ~(size_t)3) + 4 * wave_floats(k)) * sizeof(float); };
    while (imgw > 1 && lds_bytes(imgw) > 78 * 1024) --imgw;  // two workgroups per CU
    imgw = pick_leaves_per_wave(B, PIX, imgw);
    const int nt = (imgw * PIX + 15) / 16;
    const size_t wf = wave_floats(imgw), lds = lds_bytes(imgw);
    if (lds > ctx->lds_per_cu) return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: a %dx%d image needs %zu bytes of LDS per workgroup, the device has %zu per CU", H, W, lds, ctx->lds_per_cu);
    const long long tasks = (B + imgw - 1) / imgw;
    static const int cp_wgs = getenv("RP_CONVPOOL_WGS") ? atoi(getenv("RP_CONVPOOL_WGS")) : 2;  // resident workgroups per CU (experiments)
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)cp_wgs, ctx->lds_per_cu / lds));
    const dim3 grid((unsigned)std::min<long long>((tasks + 3) / 4, (long long)ctx->n_cu * per_cu)), block(256);  // persistent waves
#define CP_LAUNCH(NT_, CIN_)                                                                                                                         \
    {                                                                                                                                                \
        const int rc_ = allow_lds(ctx, (const void *)k_convpool32<NT_, CIN_>, lds, "rp_nn_convpool32"); if (rc_ != RP_OK) return rc_;                \
        hipLaunchKernelGGL((k_convpool32<NT_, CIN_>), grid, block, lds, ctx->stream, x_dev, frag_dev, bias_dev, out_dev, (long long)B, (int)H,        \
                           (int)W, imgw, (int)wf, ctx->nn_rows_dev);                                                                                 \
    }
    static const int cp_tail = getenv("RP_CONVPOOL_TAIL") ? atoi(getenv("RP_CONVPOOL_TAIL")) : 1;  // 0: the tail pixels as a whole tile (A/B)
    const int tail_px = (imgw * PIX) & 15;
    if (cp_tail && Cin == 16 && nt == 7 && tail_px >= 1 && tail_px <= 4) {  // six tiles + a tail of up to four pixels (10x10: 96 + 4)
        const int rc_ = allow_lds(ctx, (const void *)k_convpool32<6, 16, true>, lds, "rp_nn_convpool32"); if (rc_ != RP_OK) return rc_;
        hipLaunchKernelGGL((k_convpool32<6, 16, true>), grid, block, lds, ctx->stream, x_dev, frag_dev, bias_dev, out_dev, (long long)B, (int)H, (int)W, imgw, (int)wf, ctx->nn_rows_dev);
    } else if (Cin == 16) {
        switch (nt) {
            case 1: CP_LAUNCH(1, 16) break; case 2: CP_LAUNCH(2, 16) break; case 3: CP_LAUNCH(3, 16) break; case 4: CP_LAUNCH(4, 16) break;
            case 5: CP_LAUNCH(5, 16) break; case 6: CP_LAUNCH(6, 16) break; case 7: CP_LAUNCH(7, 16) break;
            default: return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: unsupported image size");
        }
    } else {
        switch (nt) {
            case 1: CP_LAUNCH(1, 32) break; case 2: CP_LAUNCH(2, 32) break; case 3: CP_LAUNCH(3, 32) break; case 4: CP_LAUNCH(4, 32) break;
            case 5: CP_LAUNCH(5, 32) break;
            default: return fail(ctx, RP_ERR_ARG, "rp_nn_convpool32: unsupported image size");
        }
    }
#undef CP_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_nn_resstage32(rp_ctx *ctx, const float *x_dev, const float *frag4_dev, const float *bias4_dev, float *out_dev, float *out_relu_dev, int64_t B,
                                int32_t H, int32_t W) {
    if (!ctx || !x_dev || !frag4_dev || !bias4_dev || !out_dev || B < 0 || H < 1 || W < 1 || H * W > 512)
        return fail(ctx, RP_ERR_ARG, "rp_nn_resstage32: bad argument (images of at most 512 pixels)");
    if (B == 0) return RP_OK;
    const int PIX = H * W;
    if (PIX > 80) {  // IMGW leaves per workgroup (k_resstage32_wg): the group size with the best fill of 4 waves x nt tiles x 16 rows
        const size_t img_px = (size_t)r32_imgp(H, W);
        auto lds_of = [&](int k) { return (128 + (k * img_px + 1) * r32_ps(32)) * sizeof(float); };
        int imgw = 1, nt = 0, waves = 4;
        long long best = -1;
        for (int k = 1; k * PIX <= 512 && lds_of(k) <= ctx->lds_per_cu; ++k) {  // <= 32 tiles: 8 waves x 4 tiles
            const int tk = (k * PIX + 15) / 16, wk = tk > 16 ? 8 : 4, ntk = (tk + wk - 1) / wk;
            const long long cost = wg_group_cost(B, k, wk * ntk, ctx->n_cu);
            if (best < 0 || cost < best) { best = cost; imgw = k; nt = ntk; waves = wk; }
        }
        if (nt == 0) return fail(ctx, RP_ERR_ARG, "rp_nn_resstage32: a %dx%d image does not fit LDS", H, W);
        const size_t lds = lds_of(imgw);
        const long long tasks = (B + imgw - 1) / imgw;
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, ctx->lds_per_cu / lds));
        const dim3 grid((unsigned)std::min<long long>(tasks, (long long)ctx->n_cu * per_cu)), block(64 * waves);
#define RSW_LAUNCH(NT_, WV_)                                                                                                                        \
    case NT_ * 16 + WV_: {                                                                                                                          \
        const int rc_ = allow_lds(ctx, (const void *)k_resstage32_wg<NT_, WV_>, lds, "rp_nn_resstage32"); if (rc_ != RP_OK) return rc_;            \
        hipLaunchKernelGGL((k_resstage32_wg<NT_, WV_>), grid, block, lds, ctx->stream, x_dev, frag4_dev, bias4_dev, out_dev, out_relu_dev, (long long)B, (int)H, (int)W, imgw, ctx->nn_rows_dev); \
    } break;
        switch (nt * 16 + waves) {
            RSW_LAUNCH(2, 4) RSW_LAUNCH(3, 4) RSW_LAUNCH(4, 4) RSW_LAUNCH(3, 8) RSW_LAUNCH(4, 8)
            default: return fail(ctx, RP_ERR_ARG, "rp_nn_resstage32: unsupported image size");
        }
#undef RSW_LAUNCH
        HIPCHK(ctx, hipGetLastError());
        return RP_OK;
    }
    const size_t img_pixels = (size_t)r32_imgp(H, W);
    auto lds_bytes = [&](int k) { return ((size_t)16 * ((k * PIX + 15) / 16) + 128 + 4 * (k * img_pixels + 1) * r32_ps(32)) * sizeof(float); };
    int imgw_max = 80 / PIX;                                       // leaves per wave: at most 5 pixel tiles of 16
    while (imgw_max > 1 && lds_bytes(imgw_max) > 78 * 1024) --imgw_max;  // two workgroups per CU
    const int imgw = pick_leaves_per_wave(B, PIX, imgw_max);
    const int nt = (imgw * PIX + 15) / 16;
    const size_t lds = lds_bytes(imgw);
    if (lds > ctx->lds_per_cu) return fail(ctx, RP_ERR_ARG, "rp_nn_resstage32: a %dx%d image needs %zu bytes of LDS per workgroup, the device has %zu per CU", H, W, lds, ctx->lds_per_cu);
    const long long tasks = (B + imgw - 1) / imgw;
    static const int rs32_wgs = getenv("RP_STAGE32_WGS") ? atoi(getenv("RP_STAGE32_WGS")) : 2;  // resident workgroups per CU (experiments)
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)rs32_wgs, ctx->lds_per_cu / lds));
    const dim3 grid((unsigned)std::min<long long>((tasks + 3) / 4, (long long)ctx->n_cu * per_cu)), block(256);  // persistent waves
#define RS_LAUNCH(NT_)                                                                                                                              \
    case NT_:                                                                                                                                       \
        { const int rc_ = allow_lds(ctx, (const void *)k_resstage32<NT_>, lds, "rp_nn_resstage32"); if (rc_ != RP_OK) return rc_; }                \
        hipLaunchKernelGGL(k_resstage32<NT_>, grid, block, lds, ctx->stream, x_dev, frag4_dev, bias4_dev, out_dev, out_relu_dev,                    \
                           (long long)B, (int)H, (int)W, imgw, ctx->nn_rows_dev);                                                                   \
        break;
    switch (nt) {
        RS_LAUNCH(1) RS_LAUNCH(2) RS_LAUNCH(3) RS_LAUNCH(4) RS_LAUNCH(5)
        default: return fail(ctx, RP_ERR_ARG, "rp_nn_resstage32: unsupported image size");
    }
#undef RS_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return RP_OK;
}

extern "C" int rp_leaf_states(rp_ctx *ctx, int32_t max_rows, uint64_t *rows_out, uint8_t *remaining_out, int32_t *slot_out, int32_t *n_out) {
    if (!ctx || max_rows < 0 || !rows_out || !remaining_out || !slot_out || !n_out) return fail(ctx, RP_ERR_ARG, "rp_leaf_states: bad argument");
    const DP &d = ctx->d;
    if (d.rows_identity) return fail(ctx, RP_ERR_STATE, "rp_leaf_states: call rp_search_step with n_leaves_out first (compact rows)");
    int n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, d.eval_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    n = std::min(n, (int)max_rows);
    *n_out = n;
    if (n == 0) return RP_OK;
    Scratch s(ctx);
    u64 *drows = s.up((const u64 *)nullptr, (size_t)n * d.H); NEED(drows);
    u8 *drem = s.up((const u8 *)nullptr, (size_t)n * d.N); NEED(drem);
    int *dslot = s.up((const int *)nullptr, (size_t)n); NEED(dslot);
    DISPATCH(ctx, k_leaf_states, grid_for(n), d, drows, drem, dslot, n);
    HIPCHK(ctx, hipMemcpyAsync(rows_out, drows, (size_t)n * d.H * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(remaining_out, drem, (size_t)n * d.N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(slot_out, dslot, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

static int launch_commit(rp_ctx *ctx, const float *pi_dev, const float *v_dev, bool logits) {
    const DP &d = ctx->d;
    const dim3 grid(grid_for(d.G)), block(64 * WAVES_PER_BLOCK);  // one wave per slot; waiting slots find their evaluator row in game_row
    const size_t lds = (size_t)WAVES_PER_BLOCK * ((size_t)d.n_leaves * sizeof(double) + (size_t)std::min(d.A, TERM_CHUNK) * sizeof(float) +
                                                 (logits ? (size_t)d.A * sizeof(float) : 0) + (size_t)(((d.A + 31) >> 5) + 1) * sizeof(u32));
    if (logits) {
        if (ctx->row64) hipLaunchKernelGGL((k_commit<u64, true>), grid, block, lds, ctx->stream, d, pi_dev, v_dev);
        else hipLaunchKernelGGL((k_commit<u32, true>), grid, block, lds, ctx->stream, d, pi_dev, v_dev);
    } else {
        if (ctx->row64) hipLaunchKernelGGL((k_commit<u64, false>), grid, block, lds, ctx->stream, d, pi_dev, v_dev);
        else hipLaunchKernelGGL((k_commit<u32, false>), grid, block, lds, ctx->stream, d, pi_dev, v_dev);
    }
    hipError_t le_ = hipGetLastError();
    if (le_ != hipSuccess) return fail(ctx, RP_ERR_DEVICE, "launch of k_commit failed: %s", hipGetErrorString(le_));
    return RP_OK;
}

extern "C" int rp_commit_eval(rp_ctx *ctx, const float *pi_dev, const float *v_dev) {
    if (!ctx || !pi_dev || !v_dev) return fail(ctx, RP_ERR_ARG, "rp_commit_eval: bad argument");
    return launch_commit(ctx, pi_dev, v_dev, false);
}

extern "C" int rp_commit_eval_logits(rp_ctx *ctx, const float *logits_dev, const float *v_dev) {
    if (!ctx || !logits_dev || !v_dev) return fail(ctx, RP_ERR_ARG, "rp_commit_eval_logits: bad argument");
    if ((size_t)WAVES_PER_BLOCK * ctx->d.A * sizeof(float) > 24 * 1024)
        return fail(ctx, RP_ERR_ARG, "rp_commit_eval_logits: %d actions do not fit the kernel's LDS row buffers (at most 1536): take the softmax first and call rp_commit_eval", ctx->d.A);
    return launch_commit(ctx, logits_dev, v_dev, true);
}

extern "C" int rp_commit_eval_host(rp_ctx *ctx, const float *pi_host, const float *v_host, int32_t n_rows) {
    if (!ctx || !pi_host || !v_host || n_rows < 0) return fail(ctx, RP_ERR_ARG, "rp_commit_eval_host: bad argument");
    const DP &d = ctx->d;
    if (d.rows_identity) return fail(ctx, RP_ERR_STATE, "rp_commit_eval_host: call rp_search_step with n_leaves_out first (compact rows)");
    int n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, d.eval_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_rows < n) return fail(ctx, RP_ERR_ARG, "rp_commit_eval_host: %d rows given, %d leaves waiting", n_rows, n);
    if (n == 0) return RP_OK;
    Scratch s(ctx);
    float *dpi = s.up(pi_host, (size_t)n * d.A); NEED(dpi);
    float *dv = s.up(v_host, (size_t)n); NEED(dv);
    { const int rc = launch_commit(ctx, dpi, dv, false); if (rc != RP_OK) return rc; }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_device_error(ctx);
}

extern "C" int rp_root_counts(rp_ctx *ctx, int32_t first, int32_t count, uint32_t *counts_out) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G || !counts_out) return fail(ctx, RP_ERR_ARG, "rp_root_counts: bad argument");
    if (count == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    u32 *dc = s.up((const u32 *)nullptr, (size_t)count * d.A); NEED(dc);
    hipLaunchKernelGGL(k_root_counts, dim3(grid_for(count)), dim3(64 * WAVES_PER_BLOCK), 0, ctx->stream, d, (int)first, (int)count, dc);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(counts_out, dc, (size_t)count * d.A * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_game_status(rp_ctx *ctx, int32_t first, int32_t count, int32_t *phase_out, int32_t *sims_done_out, int32_t *moves_out,
                              uint64_t *episode_out) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G) return fail(ctx, RP_ERR_ARG, "rp_game_status: bad argument");
    const DP &d = ctx->d;
    if (phase_out) HIPCHK(ctx, hipMemcpyAsync(phase_out, d.phase + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (sims_done_out) HIPCHK(ctx, hipMemcpyAsync(sims_done_out, d.sims_done + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (moves_out) HIPCHK(ctx, hipMemcpyAsync(moves_out, d.moves + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (episode_out) HIPCHK(ctx, hipMemcpyAsync(episode_out, d.episode + first, (size_t)count * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

extern "C" int rp_advance_roots(rp_ctx *ctx, int32_t first, int32_t count, const int32_t *action, int32_t *ended_out, double *score_out) {
    if (!ctx || first < 0 || count < 0 || first + count > ctx->d.G || !action) return fail(ctx, RP_ERR_ARG, "rp_advance_roots: bad argument");
    if (count == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    int *dact = s.up(action, (size_t)count); NEED(dact);
    DISPATCH_STAGED(ctx, k_advance, grid_for(count), d, (int)first, (int)count, (const int *)dact);
    if (ended_out) HIPCHK(ctx, hipMemcpyAsync(ended_out, d.last_outcome + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (score_out) HIPCHK(ctx, hipMemcpyAsync(score_out, d.last_score + first, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return check_device_error(ctx);
}

extern "C" int rp_pop_finished(rp_ctx *ctx, int64_t max_n, uint64_t *episode_id_out, int32_t *outcome_out, double *score_out,
                               int32_t *moves_out, int64_t *n_out) {
    if (!ctx || max_n < 0 || !n_out) return fail(ctx, RP_ERR_ARG, "rp_pop_finished: bad argument");
    const DP &d = ctx->d;
    int cnt = 0;
    HIPCHK(ctx, hipMemcpyAsync(&cnt, d.fin_count, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    cnt = std::min(cnt, d.fin_cap);
    int64_t avail = cnt - ctx->fin_popped, n = std::min(avail, max_n);
    *n_out = n;
    if (n <= 0) { *n_out = 0; return RP_OK; }
    size_t off = (size_t)ctx->fin_popped;
    if (episode_id_out) HIPCHK(ctx, hipMemcpyAsync(episode_id_out, d.fin_episode + off, n * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    if (outcome_out) HIPCHK(ctx, hipMemcpyAsync(outcome_out, d.fin_outcome + off, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (score_out) HIPCHK(ctx, hipMemcpyAsync(score_out, d.fin_score + off, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (moves_out) HIPCHK(ctx, hipMemcpyAsync(moves_out, d.fin_moves + off, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->fin_popped += n;
    if (ctx->fin_popped == cnt) {  // ring drained: rewind
        HIPCHK(ctx, hipMemsetAsync(d.fin_count, 0, sizeof(int), ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->fin_popped = 0;
    }
    return RP_OK;
}

extern "C" int rp_set_instance_pool(rp_ctx *ctx, int64_t n_instances, const uint8_t *item_wh, const int32_t *total_area, uint64_t first_id) {
    if (!ctx || n_instances < 0 || (n_instances > 0 && (!item_wh || !total_area))) return fail(ctx, RP_ERR_ARG, "rp_set_instance_pool: bad argument");
    DP &d = ctx->d;
    std::vector<int> mh((size_t)n_instances, 0);
    for (int64_t k = 0; k < n_instances; ++k)
        for (int i = 0; i < d.N; ++i) {
            int w = item_wh[((size_t)k * d.N + i) * 2], h = item_wh[((size_t)k * d.N + i) * 2 + 1];
            if (w < 1 || w > d.W || h < 1 || h > d.H) return fail(ctx, RP_ERR_ARG, "item %d of instance %lld has size %dx%d outside the %dx%d grid", i, (long long)k, w, h, d.W, d.H);
            mh[k] = std::max(mh[k], h);
        }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_instances > ctx->pool_cap) {  // grow (old buffers stay in ctx->allocs until destroy)
        ALLOC(ctx, ctx->pool_wh, (size_t)n_instances * d.N * 2);
        ALLOC(ctx, ctx->pool_area, (size_t)n_instances);
        ALLOC(ctx, ctx->pool_max_h, (size_t)n_instances);
        ctx->pool_cap = n_instances;
    }
    if (n_instances > 0) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->pool_wh, item_wh, (size_t)n_instances * d.N * 2, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pool_area, total_area, (size_t)n_instances * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->pool_max_h, mh.data(), (size_t)n_instances * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipMemsetAsync(d.next_instance, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    d.pool_wh = ctx->pool_wh; d.pool_area = ctx->pool_area; d.pool_max_h = ctx->pool_max_h;
    d.n_instances = n_instances; d.first_id = first_id;
    const PoolDesc pd = {(long long)n_instances, (unsigned long long)first_id, ctx->pool_wh, ctx->pool_area, ctx->pool_max_h};
    HIPCHK(ctx, hipMemcpy((void *)d.pool_desc, &pd, sizeof pd, hipMemcpyHostToDevice));
    return RP_OK;
}

// instances from generator seeds into a DEVICE buffer [n][N][2] (shared by rp_generate_items and rp_set_instance_pool_seeds)
static int generate_items_dev(rp_ctx *ctx, int64_t n, const uint32_t *seeds_host, int bin_w, int bin_h, u8 *out_dev) {
    const DP &d = ctx->d;
    if (bin_w < 1 || bin_w > d.W || bin_h < 1 || bin_h > d.H || bin_w > 255 || bin_h > 255)
        return fail(ctx, RP_ERR_ARG, "generator rectangle %dx%d does not fit the %dx%d grid", bin_w, bin_h, d.W, d.H);
    if ((int64_t)bin_w * bin_h < d.N) return fail(ctx, RP_ERR_ARG, "a %dx%d rectangle cannot be cut into %d items", bin_w, bin_h, d.N);
    Scratch s(ctx);
    u32 *dseeds = s.up((const u32 *)seeds_host, (size_t)n); NEED(dseeds);
    u32 *dmt = s.up((const u32 *)nullptr, (size_t)n * 624); NEED(dmt);
    hipLaunchKernelGGL(k_items_generator, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (long long)n, d.N, bin_w, bin_h, (const u32 *)dseeds, dmt, out_dev);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // scratch is freed on return
    return RP_OK;
}

extern "C" int rp_generate_items(rp_ctx *ctx, int64_t n, const uint32_t *seeds, int32_t bin_w, int32_t bin_h, uint8_t *item_wh_out) {
    if (!ctx || n < 0 || !seeds || !item_wh_out) return fail(ctx, RP_ERR_ARG, "rp_generate_items: bad argument");
    if (n == 0) return RP_OK;
    Scratch s(ctx);
    u8 *dout = s.up((const u8 *)nullptr, (size_t)n * ctx->d.N * 2); NEED(dout);
    int rc = generate_items_dev(ctx, n, seeds, bin_w, bin_h, dout);
    if (rc != RP_OK) return rc;
    HIPCHK(ctx, hipMemcpy(item_wh_out, dout, (size_t)n * ctx->d.N * 2, hipMemcpyDeviceToHost));
    return RP_OK;
}

extern "C" int rp_set_instance_pool_seeds(rp_ctx *ctx, int64_t n_instances, const uint32_t *seeds, int32_t bin_w, int32_t bin_h, uint64_t first_id) {
    if (!ctx || n_instances < 0 || (n_instances > 0 && !seeds)) return fail(ctx, RP_ERR_ARG, "rp_set_instance_pool_seeds: bad argument");
    DP &d = ctx->d;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n_instances > ctx->pool_cap) {
        ALLOC(ctx, ctx->pool_wh, (size_t)n_instances * d.N * 2);
        ALLOC(ctx, ctx->pool_area, (size_t)n_instances);
        ALLOC(ctx, ctx->pool_max_h, (size_t)n_instances);
        ctx->pool_cap = n_instances;
    }
    if (n_instances > 0) {
        int rc = generate_items_dev(ctx, n_instances, seeds, bin_w, bin_h, ctx->pool_wh);
        if (rc != RP_OK) return rc;
        // total area = the generator rectangle (CoachBPP.py:119); max_h over all items (BinPackingGame.py:41-50)
        std::vector<u8> wh((size_t)n_instances * d.N * 2);
        HIPCHK(ctx, hipMemcpy(wh.data(), ctx->pool_wh, wh.size(), hipMemcpyDeviceToHost));
        std::vector<int> area((size_t)n_instances, bin_w * bin_h), mh((size_t)n_instances, 0);
        for (int64_t k = 0; k < n_instances; ++k)
            for (int i = 0; i < d.N; ++i) mh[k] = std::max(mh[k], (int)wh[((size_t)k * d.N + i) * 2 + 1]);
        HIPCHK(ctx, hipMemcpy(ctx->pool_area, area.data(), area.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->pool_max_h, mh.data(), mh.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    HIPCHK(ctx, hipMemsetAsync(d.next_instance, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    d.pool_wh = ctx->pool_wh; d.pool_area = ctx->pool_area; d.pool_max_h = ctx->pool_max_h;
    d.n_instances = n_instances; d.first_id = first_id;
    const PoolDesc pd = {(long long)n_instances, (unsigned long long)first_id, ctx->pool_wh, ctx->pool_area, ctx->pool_max_h};
    HIPCHK(ctx, hipMemcpy((void *)d.pool_desc, &pd, sizeof pd, hipMemcpyHostToDevice));
    return RP_OK;
}

extern "C" int rp_begin_pool(rp_ctx *ctx) {
    if (!ctx) return RP_ERR_ARG;
    const DP &d = ctx->d;
    if (!d.pool_wh) return fail(ctx, RP_ERR_STATE, "rp_begin_pool: no instance pool set");
    DISPATCH_STAGED(ctx, k_pool_begin, grid_for(d.G), d);
    return check_device_error(ctx);
}

extern "C" int rp_examples_count(rp_ctx *ctx, int64_t *n_out) {
    if (!ctx || !n_out) return fail(ctx, RP_ERR_ARG, "rp_examples_count: bad argument");
    unsigned long long n = 0;
    HIPCHK(ctx, hipMemcpyAsync(&n, ctx->d.ex_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = std::min<int64_t>((int64_t)n, ctx->d.max_examples);
    return RP_OK;
}

static PackedEx own_examples(const DP &d) {
    PackedEx ex;
    ex.key = d.ex_key; ex.wh = d.ex_wh; ex.value = d.ex_value; ex.sp_off64 = nullptr; ex.sp_off32 = d.ex_sp_off; ex.sp_n = (const int *)d.ex_sp_n;
    ex.sp_act = d.ex_sp_act; ex.sp_cnt = d.ex_sp_cnt; ex.n_examples = d.max_examples; ex.n_sparse = d.sp_cap;
    return ex;
}

extern "C" int rp_examples_tensors(rp_ctx *ctx, int64_t first, int64_t count, float *planes_dev, float *pi_dev, float *value_dev) {
    if (!ctx || first < 0 || count < 0 || !planes_dev || !pi_dev || !value_dev) return fail(ctx, RP_ERR_ARG, "rp_examples_tensors: bad argument");
    int64_t n = 0;
    int rc = rp_examples_count(ctx, &n);
    if (rc != RP_OK) return rc;
    if (first + count > n) return fail(ctx, RP_ERR_ARG, "rp_examples_tensors: range [%lld,%lld) exceeds the %lld recorded examples", (long long)first, (long long)(first + count), (long long)n);
    if (count == 0) return RP_OK;
    const DP &d = ctx->d;
    DISPATCH(ctx, k_examples, grid_for(count), d, own_examples(d), (long long)first, (long long)count, (const long long *)nullptr, planes_dev, pi_dev, value_dev);
    return RP_OK;
}

extern "C" int rp_examples_packed_count(rp_ctx *ctx, int64_t *n_examples_out, int64_t *n_sparse_out) {
    if (!ctx || !n_examples_out || !n_sparse_out) return fail(ctx, RP_ERR_ARG, "rp_examples_packed_count: bad argument");
    int rc = rp_examples_count(ctx, n_examples_out);
    if (rc != RP_OK) return rc;
    unsigned long long sp = 0;
    if (ctx->d.max_examples > 0) {
        HIPCHK(ctx, hipMemcpyAsync(&sp, ctx->d.ex_sp_cursor, sizeof sp, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_sparse_out = std::min<int64_t>((int64_t)sp, ctx->d.sp_cap);
    return RP_OK;
}

extern "C" int rp_examples_packed(rp_ctx *ctx, int64_t n_examples, int64_t n_sparse, uint32_t *key_dev, uint8_t *item_wh_dev, int32_t *value_dev,
                                  int32_t *sp_off_dev, int32_t *sp_n_dev, uint16_t *sp_act_dev, uint32_t *sp_cnt_dev, int64_t *episode_dev, int32_t *move_dev) {
    if (!ctx || n_examples < 0 || n_sparse < 0 || (n_examples > 0 && (!key_dev || !item_wh_dev || !value_dev || !sp_off_dev || !sp_n_dev)) ||
        (n_sparse > 0 && (!sp_act_dev || !sp_cnt_dev)))
        return fail(ctx, RP_ERR_ARG, "rp_examples_packed: bad argument");
    int64_t n = 0, ns = 0;
    int rc = rp_examples_packed_count(ctx, &n, &ns);
    if (rc != RP_OK) return rc;
    if (n_examples > n || n_sparse > ns) return fail(ctx, RP_ERR_ARG, "rp_examples_packed: %lld examples / %lld pairs asked for, %lld / %lld recorded", (long long)n_examples, (long long)n_sparse, (long long)n, (long long)ns);
    const DP &d = ctx->d;
    const size_t E = (size_t)n_examples;
    if (E) {
        HIPCHK(ctx, hipMemcpyAsync(key_dev, d.ex_key, E * d.KW * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(item_wh_dev, d.ex_wh, E * d.N * 2, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(value_dev, d.ex_value, E * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(sp_off_dev, d.ex_sp_off, E * 4, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(sp_n_dev, d.ex_sp_n, E * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (episode_dev) HIPCHK(ctx, hipMemcpyAsync(episode_dev, d.ex_episode, E * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (move_dev) HIPCHK(ctx, hipMemcpyAsync(move_dev, d.ex_move, E * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (n_sparse) {
        HIPCHK(ctx, hipMemcpyAsync(sp_act_dev, d.ex_sp_act, (size_t)n_sparse * 2, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(sp_cnt_dev, d.ex_sp_cnt, (size_t)n_sparse * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return RP_OK;
}

extern "C" int rp_expand_examples(rp_ctx *ctx, int64_t n, const int64_t *index_dev, int64_t n_examples, int64_t n_sparse, const uint32_t *key_dev,
                                  const uint8_t *item_wh_dev, const int32_t *value_dev, const int64_t *sp_off_dev, const int32_t *sp_n_dev,
                                  const uint16_t *sp_act_dev, const uint32_t *sp_cnt_dev, float *planes_dev, float *pi_dev, float *value_out_dev) {
    if (!ctx || n < 0 || n_examples < 0 || n_sparse < 0 || !planes_dev || !pi_dev || !value_out_dev ||
        (n_examples > 0 && (!key_dev || !item_wh_dev || !value_dev || !sp_off_dev || !sp_n_dev)) || (n_sparse > 0 && (!sp_act_dev || !sp_cnt_dev)))
        return fail(ctx, RP_ERR_ARG, "rp_expand_examples: bad argument");
    if (n == 0) return RP_OK;
    if (!index_dev && n > n_examples) return fail(ctx, RP_ERR_ARG, "rp_expand_examples: %lld rows asked of %lld examples", (long long)n, (long long)n_examples);
    PackedEx ex;
    ex.key = key_dev; ex.wh = item_wh_dev; ex.value = value_dev; ex.sp_off64 = (const long long *)sp_off_dev; ex.sp_off32 = nullptr; ex.sp_n = sp_n_dev;
    ex.sp_act = sp_act_dev; ex.sp_cnt = sp_cnt_dev; ex.n_examples = n_examples; ex.n_sparse = n_sparse;
    DISPATCH(ctx, k_examples, grid_for(n), ctx->d, ex, 0LL, (long long)n, (const long long *)index_dev, planes_dev, pi_dev, value_out_dev);
    return RP_OK;
}

extern "C" int rp_check(rp_ctx *ctx) {
    if (!ctx) return RP_ERR_ARG;
    return check_device_error(ctx);
}

extern "C" int rp_examples_meta(rp_ctx *ctx, int64_t first, int64_t count, uint64_t *episode_id_out, int32_t *move_out) {
    if (!ctx || first < 0 || count < 0 || !episode_id_out || !move_out) return fail(ctx, RP_ERR_ARG, "rp_examples_meta: bad argument");
    int64_t n = 0;
    int rc = rp_examples_count(ctx, &n);
    if (rc != RP_OK) return rc;
    if (first + count > n) return fail(ctx, RP_ERR_ARG, "rp_examples_meta: range [%lld,%lld) exceeds the %lld recorded examples", (long long)first, (long long)(first + count), (long long)n);
    if (count == 0) return RP_OK;
    HIPCHK(ctx, hipMemcpyAsync(episode_id_out, ctx->d.ex_episode + first, (size_t)count * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(move_out, ctx->d.ex_move + first, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_examples_clear(rp_ctx *ctx) {
    if (!ctx) return RP_ERR_ARG;
    HIPCHK(ctx, hipMemsetAsync(ctx->d.ex_count, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d.ex_sp_cursor, 0, sizeof(unsigned long long), ctx->stream));
    return RP_OK;
}

extern "C" int rp_counters(rp_ctx *ctx, int64_t *out16, int32_t reset) {
    if (!ctx || !out16) return fail(ctx, RP_ERR_ARG, "rp_counters: bad argument");
    hipLaunchKernelGGL(k_reduce_counters, dim3(1), dim3(1024), 0, ctx->stream, ctx->d);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out16, ctx->d.counters, CNT_N * sizeof(u64), hipMemcpyDeviceToHost, ctx->stream));
    if (reset) HIPCHK(ctx, hipMemsetAsync(ctx->d.slot_cnt, 0, (size_t)ctx->d.G * CNT_N * sizeof(u64), ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_tree_size(rp_ctx *ctx, int32_t slot, int32_t *n_nodes_out, int32_t *n_edges_out) {
    if (!ctx || slot < 0 || slot >= ctx->d.G || !n_nodes_out || !n_edges_out) return fail(ctx, RP_ERR_ARG, "rp_tree_size: bad argument");
    HIPCHK(ctx, hipMemcpyAsync(n_nodes_out, ctx->d.n_nodes + slot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_edges_out = ctx->d.edge_cap;  // legal-move runs live in level chunks: the dense edge view spans the slot's whole arena
    return RP_OK;
}

extern "C" int rp_arena_peak(rp_ctx *ctx, int32_t *prior_chunks_out, int32_t *visited_chunks_out, int32_t *chunk_entries_out2) {
    if (!ctx || !prior_chunks_out || !visited_chunks_out) return fail(ctx, RP_ERR_ARG, "rp_arena_peak: bad argument");
    u32 pk[2] = {0, 0};
    u32 *tail = ctx->d.peak_chunks + (size_t)ctx->d.G * 2;
    hipLaunchKernelGGL(k_reduce_peaks, dim3(1), dim3(256), 0, ctx->stream, ctx->d, tail);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(pk, tail, sizeof pk, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *prior_chunks_out = (int)pk[0]; *visited_chunks_out = (int)pk[1];
    if (chunk_entries_out2) { chunk_entries_out2[0] = ctx->d.pchunk; chunk_entries_out2[1] = ctx->d.vchunk; }
    return RP_OK;
}

extern "C" int rp_dump_tree(rp_ctx *ctx, int32_t slot, uint64_t *node_rows, uint8_t *node_remaining, int8_t *node_term, uint8_t *node_term_kind,
                            uint8_t *node_expanded, uint32_t *node_ns, uint32_t *node_edge_off, uint32_t *node_n_valid, uint16_t *edge_action,
                            double *edge_p, double *edge_q, uint32_t *edge_nsa, uint8_t *edge_q_kind, uint32_t *edge_child) {
    if (!ctx || slot < 0 || slot >= ctx->d.G) return fail(ctx, RP_ERR_ARG, "rp_dump_tree: bad argument");
    const DP &d = ctx->d;
    int nn = 0, ne = 0;
    int rc = rp_tree_size(ctx, slot, &nn, &ne);
    if (rc != RP_OK) return rc;
    const int nv = d.vis_cap;
    std::vector<NodeHdr> hdr(nn);
    std::vector<u32> key((size_t)nn * d.KW);
    std::vector<VisEntry> vis(nv);
    std::vector<float> pi(ne);

    HIPCHK(ctx, hipMemcpy(hdr.data(), slot_region(d, d.hdr, slot), (size_t)nn * sizeof(NodeHdr), hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(key.data(), slot_region(d, d.key, slot), (size_t)nn * d.KW * 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(edge_action, slot_region(d, d.pAct, slot), (size_t)ne * 2, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(pi.data(), slot_region(d, d.pPi, slot), (size_t)ne * 4, hipMemcpyDeviceToHost));
    HIPCHK(ctx, hipMemcpy(vis.data(), slot_region(d, d.vis, slot), (size_t)nv * sizeof(VisEntry), hipMemcpyDeviceToHost));
    for (int i = 0; i < nn; ++i) {
        const NodeHdr &h = hdr[i];
        const u32 *k = key.data() + (size_t)i * d.KW;
        for (int r = 0; r < d.H; ++r) node_rows[(size_t)i * d.H + r] = ctx->row64 ? ((const u64 *)k)[r] : (u64)k[r];
        const u32 *rw = k + d.H * d.RW;
        for (int it = 0; it < d.N; ++it) node_remaining[(size_t)i * d.N + it] = (u8)((rw[it >> 5] >> (it & 31)) & 1u);
        node_term[i] = h.term; node_term_kind[i] = (u8)hdr_term_kind(h); node_expanded[i] = (h.flags & HF_EXPANDED) ? 1 : 0;
        node_ns[i] = h.ns; node_edge_off[i] = h.prior_off; node_n_valid[i] = h.n_valid;
        // dense per-legal-move view: the prior as the search would compute it, statistics only where an edge was visited
        if ((size_t)h.prior_off + h.n_valid > (size_t)ne || (h.vis_n && (size_t)h.vis_off + h.vis_n > (size_t)nv))
            return fail(ctx, RP_ERR_STATE, "rp_dump_tree: node %d points outside its arena", i);
        for (u32 e = h.prior_off; e < h.prior_off + h.n_valid; ++e) {
            double x = (double)pi[e] * 1.0;
            edge_p[e] = !(h.flags & HF_EXPANDED) ? 0.0 : ((h.flags & HF_FALLBACK) ? (x + 1.0) / h.norm : x / h.norm);
            edge_q[e] = 0.0; edge_nsa[e] = 0; edge_q_kind[e] = 0; edge_child[e] = NONE32;
        }
        for (u32 j = 0; j < h.vis_n; ++j) {
            u32 v = h.vis_off + j, e = h.prior_off + vis[v].idx;
            if (edge_p[e] != vis[v].p) return fail(ctx, RP_ERR_STATE, "rp_dump_tree: stored prior of a visited edge differs from pi / norm");
            edge_q[e] = vis[v].q; edge_nsa[e] = vis[v].n & NSA_MASK; edge_q_kind[e] = (u8)(vis[v].n >> 30); edge_child[e] = vis[v].child;
        }
    }
    return RP_OK;
}

extern "C" int rp_selftest_sqrt(rp_ctx *ctx, int64_t n, double *sqrt_n_out, double *sqrt_n_eps_out) {
    if (!ctx || n < 0 || !sqrt_n_out || !sqrt_n_eps_out) return fail(ctx, RP_ERR_ARG, "rp_selftest_sqrt: bad argument");
    if (n == 0) return RP_OK;
    Scratch s(ctx);
    double *a = s.up((const double *)nullptr, (size_t)n); NEED(a);
    double *b = s.up((const double *)nullptr, (size_t)n); NEED(b);
    hipLaunchKernelGGL(k_selftest_sqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, a, b);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(sqrt_n_out, a, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(sqrt_n_eps_out, b, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_selftest_q_update(rp_ctx *ctx, int64_t n, const double *q, const uint8_t *q_kind, const uint32_t *nsa, const double *v,
                                    const uint8_t *v_kind, double *q_out, uint8_t *q_kind_out) {
    if (!ctx || n < 0 || !q || !q_kind || !nsa || !v || !v_kind || !q_out || !q_kind_out) return fail(ctx, RP_ERR_ARG, "rp_selftest_q_update: bad argument");
    if (n == 0) return RP_OK;
    Scratch s(ctx);
    double *dq = s.up(q, (size_t)n); NEED(dq);
    u8 *dqk = s.up(q_kind, (size_t)n); NEED(dqk);
    u32 *dn = s.up(nsa, (size_t)n); NEED(dn);
    double *dv = s.up(v, (size_t)n); NEED(dv);
    u8 *dvk = s.up(v_kind, (size_t)n); NEED(dvk);
    double *dqo = s.up((const double *)nullptr, (size_t)n); NEED(dqo);
    u8 *dqko = s.up((const u8 *)nullptr, (size_t)n); NEED(dqko);
    hipLaunchKernelGGL(k_selftest_q, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (long long)n, dq, dqk, dn, dv, dvk, dqo, dqko);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(q_out, dqo, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(q_kind_out, dqko, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}

extern "C" int rp_selftest_masked_prior(rp_ctx *ctx, int64_t B, const float *pi, const uint8_t *valid, double *p_out) {
    if (!ctx || B < 0 || !pi || !valid || !p_out) return fail(ctx, RP_ERR_ARG, "rp_selftest_masked_prior: bad argument");
    if (B == 0) return RP_OK;
    const DP &d = ctx->d;
    Scratch s(ctx);
    float *dpi = s.up(pi, (size_t)B * d.A); NEED(dpi);
    u8 *dva = s.up(valid, (size_t)B * d.A); NEED(dva);
    double *dout = s.up((const double *)nullptr, (size_t)B * d.A); NEED(dout);
    u16 *dact = s.up((const u16 *)nullptr, (size_t)B * d.A); NEED(dact);
    float *dpc = s.up((const float *)nullptr, (size_t)B * d.A); NEED(dpc);
    hipLaunchKernelGGL(k_selftest_prior, dim3(grid_for(B)), dim3(64 * WAVES_PER_BLOCK), 0, ctx->stream, d, (long long)B, (const float *)dpi,
                       (const u8 *)dva, dout, dact, dpc);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(p_out, dout, (size_t)B * d.A * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RP_OK;
}
