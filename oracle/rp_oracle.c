/*
 * rp_oracle.c -- CPU restatement of the reference's self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see rp_oracle.h).  Plain C, one byte per grid
 * cell, written to mirror the reference statement by statement; it is the
 * checker for the HIP engine and the "port" CPU baseline of bench.py.
 * Build with -ffp-contract=off: every f32/f64 operation below must round
 * separately, exactly as the NumPy / CPython expressions it restates.
 *
 * All paths below are relative to /root/reference/xw_mcts.
 */
#include "rp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_version(void) { return 1; }

/* ------------------------------------------------------------------------- */
/* np.sum(float64[n]) -- NumPy's pairwise summation (numpy/_core/src/umath/
 * loops_utils.h.src, DOUBLE_pairwise_sum, contiguous case): blocks of <=128
 * elements are summed with 8 strided accumulators, larger ranges are split at
 * n/2 rounded down to a multiple of 8.  MCTS_bpp.py:90,100 call np.sum on the
 * masked prior, so the renormalised P depends on this order bit for bit. */
static double pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}
double orc_np_sum_f64(const double *a, int64_t n) { return pairwise_sum(a, n); }

/* ------------------------------------------------------------------------- */
/* sum(board[r, j:j+w]) with NumPy slice clipping at W */
static int row_window_sum(int W, const uint8_t *board, int r, int j, int w) {
    int s = 0, hi = j + w < W ? j + w : W;
    for (int c = j; c < hi; c++) s += board[r * W + c];
    return s;
}

/* Bin.get_adjacency (BinPackingLogic.py:47-78): only the "left" rule is live.
 * j == 0 counts as adjacent; otherwise t = first row whose window j..j+w-1 is
 * empty (the for/break leaves t = H-1 when no row is empty, :66-68) and the
 * cell left of the window in that row must be occupied (:69). */
static int adjacency(int W, int H, const uint8_t *board, int j, int w) {
    if (j == 0) return 1;
    int t = 0;
    for (t = 0; t < H; t++)
        if (row_window_sum(W, board, t, j, w) == 0) break;
    if (t == H) t = H - 1;
    return board[t * W + (j - 1)] > 0;
}

/* Bin.get_moves_for_square (BinPackingLogic.py:80-93): for j in 0..W-w the
 * move is legal iff the occupied CELL count of columns j..j+w-1 is at most
 * w*H - w*h (:89 -- a cell count, not a free-row count) and get_adjacency. */
static int moves_for_item(int W, int H, const uint8_t *board, int w, int h, uint8_t *row_out /* W flags or NULL */) {
    int n = 0;
    for (int j = 0; j < W - w + 1; j++) {
        int occupied = 0;
        for (int r = 0; r < H; r++) occupied += row_window_sum(W, board, r, j, w);
        if (occupied <= w * H - w * h && adjacency(W, H, board, j, w)) {
            if (row_out) row_out[j] = 1;
            n++;
        }
    }
    return n;
}

/* BinPackingGame.getValidMoves (BinPackingGame.py:78-92): items whose plane is
 * all zero are skipped (:86); valids[item*W + x] = 1 (:91). */
int orc_valid_moves(int W, int H, int N, const uint8_t *board, const uint8_t *item_w, const uint8_t *item_h,
                    const uint8_t *remaining, uint8_t *valids) {
    int n = 0;
    memset(valids, 0, (size_t)W * N);
    for (int i = 0; i < N; i++) {
        if (!remaining[i]) continue;
        n += moves_for_item(W, H, board, item_w[i], item_h[i], valids + (size_t)i * W);
    }
    return n;
}

/* BinPackingGame.has_valid_moves (BinPackingGame.py:94-107): same rule, stops at
 * the first item that has a move. */
int orc_has_valid_moves(int W, int H, int N, const uint8_t *board, const uint8_t *item_w, const uint8_t *item_h,
                        const uint8_t *remaining) {
    for (int i = 0; i < N; i++) {
        if (!remaining[i]) continue;
        if (moves_for_item(W, H, board, item_w[i], item_h[i], NULL) > 0) return 1;
    }
    return 0;
}

/* BinPackingGame.getNextState (BinPackingGame.py:58-76): item = int(a/W), x = int(a%W)
 * (:67); Bin.execute_move (BinPackingLogic.py:95-109) scans rows 0..H-1 and fills
 * the window of every row whose window is empty until h rows are filled -- the rows
 * need not be contiguous and fewer than h may get filled; the item plane is zeroed
 * (BinPackingGame.py:53-56,75). */
int orc_next_state(int W, int H, int N, uint8_t *board, const uint8_t *item_w, const uint8_t *item_h,
                   uint8_t *remaining, int action) {
    int item = action / W, x = action % W;
    if (item < 0 || item >= N || !remaining[item]) return -1;
    int w = item_w[item], h = item_h[item], t = 0;
    int hi = x + w < W ? x + w : W;
    for (int ii = 0; ii < H; ii++) {
        if (row_window_sum(W, board, ii, x, w) == 0) {
            for (int c = x; c < hi; c++) board[ii * W + c] = 1;
            t++;
            if (t == h) break;
        }
    }
    remaining[item] = 0;
    return 0;
}

/* BinPackingGame.get_minimal_bin_height (BinPackingGame.py:181-186): 1 + highest
 * occupied row; the reversed loop falls through with i = 0 on an empty grid -> 1. */
static int minimal_bin_height(int W, int H, const uint8_t *board) {
    int i;
    for (i = H - 1; i > 0; i--) {
        int s = 0;
        for (int c = 0; c < W; c++) s += board[i * W + c];
        if (s > 0) break;
    }
    return i + 1;
}

static int cmp_f64(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* BinPackingGame.getRankedReward (BinPackingGame.py:188-212). */
int orc_ranked_reward(int W, int H, const uint8_t *board, int64_t total_area, int max_h, const double *rewards,
                      int n_rewards, double alpha, double *r_out) {
    int64_t cells = 0;
    for (int i = 0; i < W * H; i++) cells += board[i];
    double r;
    if (cells != total_area) {
        r = 0.0; /* :193-195 */
    } else {
        /* :198  max(np.ceil(area / W), max_h) / get_minimal_bin_height(board) */
        double need = ceil((double)total_area / (double)W);
        double top = need >= (double)max_h ? need : (double)max_h;
        r = top / (double)minimal_bin_height(W, H, board);
    }
    if (r_out) *r_out = r;
    if (n_rewards == 0) return 1; /* :203-204 */
    double *sorted = (double *)malloc(sizeof(double) * (size_t)n_rewards);
    memcpy(sorted, rewards, sizeof(double) * (size_t)n_rewards);
    qsort(sorted, (size_t)n_rewards, sizeof(double), cmp_f64); /* :205 */
    /* :206  sorted[int(np.floor(len * alpha)) - 1]; index -1 wraps to the last element */
    int idx = (int)floor((double)n_rewards * alpha) - 1;
    if (idx < 0) idx += n_rewards;
    double bl = sorted[idx];
    free(sorted);
    if (r > bl || r == 1.0) return 1; /* :207-208 */
    if (r < bl) return -1;            /* :209-210 */
    return ORC_TIE;                   /* :211-212 np.random.choice([1,-1]) */
}

/* BinPackingGame.getGameEnded (BinPackingGame.py:109-116). */
int orc_game_ended(int W, int H, int N, const uint8_t *board, const uint8_t *item_w, const uint8_t *item_h,
                   const uint8_t *remaining, int64_t total_area, int max_h, const double *rewards, int n_rewards,
                   double alpha, double *r_out) {
    if (orc_has_valid_moves(W, H, N, board, item_w, item_h, remaining)) {
        if (r_out) *r_out = 0.0;
        return 0;
    }
    return orc_ranked_reward(W, H, board, total_area, max_h, rewards, n_rewards, alpha, r_out);
}

/* ------------------------------------------------------------------------- */
/* Backup update of one edge, MCTS_bpp.py:130-136:
 *     Qsa = (Nsa * Qsa + v) / (Nsa + 1)      or     Qsa = v   on the first visit.
 * Nsa is a Python int.  What arithmetic that is depends on the Python/NumPy
 * types of Qsa and v (NumPy 2 / NEP 50 promotion, verified in the container):
 *   v is  ORC_WEAK  a Python int   (terminal value, BinPackingGame.py:204,208,210)
 *         ORC_F32   np.float32 array of shape (1,) (NNet.predict's v, NNet.py:85)
 *         ORC_F64   np.int64       (np.random.choice of BinPackingGame.py:212)
 *   Q is  ORC_WEAK  Python int/float:  + WEAK -> f64 ops, stays WEAK
 *                                       + F32  -> N*Q in f64, cast to f32, f32 add, f32 div -> F32
 *                                       + F64  -> f64 ops -> F64
 *         ORC_F32   f32 array:          + WEAK/F32 -> f32 mul, add, div -> F32
 *                                       + F64  -> f32 mul, then f64 add and div -> F64
 *         ORC_F64   f64:                 everything in f64, stays F64
 */
void orc_q_update(double *q, int *q_kind, uint32_t n, double v, int v_kind) {
    if (n == 0) {
        *q = v;
        *q_kind = v_kind;
        return;
    }
    double dn = (double)n, dn1 = (double)(n + 1);
    if (*q_kind == ORC_WEAK) {
        double nq = dn * (*q);
        if (v_kind == ORC_F32) {
            float s = (float)nq + (float)v;
            *q = (double)(s / (float)dn1);
            *q_kind = ORC_F32;
        } else {
            *q = (nq + v) / dn1;
            *q_kind = v_kind; /* WEAK stays WEAK, F64 makes it F64 */
        }
    } else if (*q_kind == ORC_F32) {
        float nq = (float)dn * (float)(*q);
        if (v_kind == ORC_F64) {
            *q = ((double)nq + v) / dn1;
            *q_kind = ORC_F64;
        } else {
            float s = nq + (float)v;
            *q = (double)(s / (float)dn1);
        }
    } else {
        *q = (dn * (*q) + v) / dn1;
    }
}

/* ------------------------------------------------------------------------- */
/* MCTS (MCTS_bpp.py).  The six dicts Qsa/Nsa/Ns/Ps/Es/Vs (:20-26) keyed by the
 * state bytes become one node record per distinct state, found through a hash
 * map on (board cells, remaining flags) -- the same equivalence classes as
 * stringRepresentation (BinPackingGame.py:214-218), because an item plane is a
 * function of (episode items, remaining flag). */
typedef struct node {
    uint8_t *key; /* H*W board cells then N remaining flags */
    int has_es, es, es_kind;
    int expanded;
    uint32_t ns;
    uint8_t *valids; /* A */
    double *p;       /* A */
    uint32_t *nsa;   /* A */
    double *q;       /* A */
    uint8_t *q_kind; /* A */
} node;

struct orc_mcts {
    int W, H, N, A, keylen;
    double cpuct, alpha;
    orc_eval_fn eval;
    void *eval_user;
    orc_tie_fn tie;
    void *tie_user;
    uint8_t *item_w, *item_h;
    int max_h;
    int64_t total_area;
    double *rewards;
    int n_rewards;
    node **nodes;
    int64_t n_nodes, cap_nodes;
    int64_t *table; /* open addressing, -1 empty */
    int64_t table_cap;
    int64_t stats[8];
    float *pi_buf;
};

static uint64_t fnv1a(const uint8_t *p, int n) {
    uint64_t h = 1469598103934665603ULL;
    for (int i = 0; i < n; i++) {
        h ^= p[i];
        h *= 1099511628211ULL;
    }
    return h;
}

static void table_insert_raw(orc_mcts *m, int64_t id) {
    uint64_t h = fnv1a(m->nodes[id]->key, m->keylen);
    int64_t mask = m->table_cap - 1, s = (int64_t)(h & (uint64_t)mask);
    while (m->table[s] >= 0) s = (s + 1) & mask;
    m->table[s] = id;
}

static void table_grow(orc_mcts *m) {
    free(m->table);
    m->table_cap *= 2;
    m->table = (int64_t *)malloc(sizeof(int64_t) * (size_t)m->table_cap);
    for (int64_t i = 0; i < m->table_cap; i++) m->table[i] = -1;
    for (int64_t i = 0; i < m->n_nodes; i++) table_insert_raw(m, i);
}

static node *lookup(orc_mcts *m, const uint8_t *key, int create, int *created) {
    uint64_t h = fnv1a(key, m->keylen);
    int64_t mask = m->table_cap - 1, s = (int64_t)(h & (uint64_t)mask);
    while (m->table[s] >= 0) {
        node *nd = m->nodes[m->table[s]];
        if (memcmp(nd->key, key, (size_t)m->keylen) == 0) {
            if (created) *created = 0;
            return nd;
        }
        s = (s + 1) & mask;
    }
    if (!create) return NULL;
    if (m->n_nodes == m->cap_nodes) {
        m->cap_nodes *= 2;
        m->nodes = (node **)realloc(m->nodes, sizeof(node *) * (size_t)m->cap_nodes);
    }
    node *nd = (node *)calloc(1, sizeof(node));
    nd->key = (uint8_t *)malloc((size_t)m->keylen);
    memcpy(nd->key, key, (size_t)m->keylen);
    m->nodes[m->n_nodes] = nd;
    m->table[s] = m->n_nodes;
    m->n_nodes++;
    if (m->n_nodes * 2 > m->table_cap) table_grow(m);
    if (created) *created = 1;
    return nd;
}

static void free_nodes(orc_mcts *m) {
    for (int64_t i = 0; i < m->n_nodes; i++) {
        node *nd = m->nodes[i];
        free(nd->key); free(nd->valids); free(nd->p); free(nd->nsa); free(nd->q); free(nd->q_kind);
        free(nd);
    }
    m->n_nodes = 0;
    for (int64_t i = 0; i < m->table_cap; i++) m->table[i] = -1;
}

orc_mcts *orc_mcts_new(int W, int H, int N, double cpuct, double alpha, orc_eval_fn eval, void *eval_user,
                       orc_tie_fn tie, void *tie_user) {
    orc_mcts *m = (orc_mcts *)calloc(1, sizeof(orc_mcts));
    m->W = W; m->H = H; m->N = N; m->A = W * N; m->keylen = W * H + N;
    m->cpuct = cpuct; m->alpha = alpha;
    m->eval = eval; m->eval_user = eval_user; m->tie = tie; m->tie_user = tie_user;
    m->item_w = (uint8_t *)calloc((size_t)N, 1);
    m->item_h = (uint8_t *)calloc((size_t)N, 1);
    m->cap_nodes = 1024;
    m->nodes = (node **)malloc(sizeof(node *) * (size_t)m->cap_nodes);
    m->table_cap = 4096;
    m->table = (int64_t *)malloc(sizeof(int64_t) * (size_t)m->table_cap);
    for (int64_t i = 0; i < m->table_cap; i++) m->table[i] = -1;
    m->pi_buf = (float *)malloc(sizeof(float) * (size_t)m->A);
    return m;
}

void orc_mcts_free(orc_mcts *m) {
    if (!m) return;
    free_nodes(m);
    free(m->nodes); free(m->table); free(m->item_w); free(m->item_h); free(m->rewards); free(m->pi_buf);
    free(m);
}

void orc_mcts_begin_episode(orc_mcts *m, const uint8_t *item_w, const uint8_t *item_h, int64_t total_area,
                            const double *rewards, int n_rewards) {
    free_nodes(m); /* MCTS.__init__: empty dicts (CoachBPP.py:124 builds a new MCTS per episode) */
    memcpy(m->item_w, item_w, (size_t)m->N);
    memcpy(m->item_h, item_h, (size_t)m->N);
    m->max_h = 0; /* BinPackingGame.getInitItems (BinPackingGame.py:41-50): max over ALL items */
    for (int i = 0; i < m->N; i++)
        if (item_h[i] > m->max_h) m->max_h = item_h[i];
    m->total_area = total_area;
    free(m->rewards);
    m->rewards = (double *)malloc(sizeof(double) * (size_t)(n_rewards > 0 ? n_rewards : 1));
    if (n_rewards > 0) memcpy(m->rewards, rewards, sizeof(double) * (size_t)n_rewards);
    m->n_rewards = n_rewards;
    memset(m->stats, 0, sizeof(m->stats));
}

typedef struct { double v; int kind; } val_t;

/* MCTS.search (MCTS_bpp.py:56-139). key = board cells + remaining flags. */
static val_t search(orc_mcts *m, const uint8_t *key, int first_traversal) {
    const int W = m->W, H = m->H, N = m->N, A = m->A;
    const uint8_t *board = key, *remaining = key + W * H;
    int created = 0;
    node *nd = lookup(m, key, 1, &created); /* s = stringRepresentation(...) (:76) */
    if (!created && first_traversal) m->stats[6]++; /* an edge led to a state another path had created */
    if (!nd->has_es) { /* :78-79 */
        double r;
        int e = orc_game_ended(W, H, N, board, m->item_w, m->item_h, remaining, m->total_area, m->max_h, m->rewards,
                               m->n_rewards, m->alpha, &r);
        nd->es_kind = ORC_WEAK;
        if (e == ORC_TIE) {
            e = m->tie ? m->tie(m->tie_user, board, remaining) : 1;
            nd->es_kind = ORC_F64; /* np.int64 from np.random.choice */
        }
        nd->es = e;
        nd->has_es = 1;
    }
    if (nd->es != 0) { /* :81-83 terminal: the ranked outcome, no sign flip */
        m->stats[2]++;
        val_t t = {(double)nd->es, nd->es_kind};
        return t;
    }
    if (!nd->expanded) { /* :85-104 leaf */
        float v32 = 0.f;
        m->eval(m->eval_user, board, remaining, m->pi_buf, &v32); /* :87 */
        nd->valids = (uint8_t *)malloc((size_t)A);
        nd->p = (double *)malloc(sizeof(double) * (size_t)A);
        nd->nsa = (uint32_t *)calloc((size_t)A, sizeof(uint32_t));
        nd->q = (double *)calloc((size_t)A, sizeof(double));
        nd->q_kind = (uint8_t *)calloc((size_t)A, 1);
        int nv = orc_valid_moves(W, H, N, board, m->item_w, m->item_h, remaining, nd->valids); /* :88 */
        for (int a = 0; a < A; a++) nd->p[a] = (double)m->pi_buf[a] * (double)nd->valids[a]; /* :89 f32*i64 -> f64 */
        double s = pairwise_sum(nd->p, A); /* :90 */
        if (s > 0) {
            for (int a = 0; a < A; a++) nd->p[a] /= s; /* :92 */
        } else { /* :93-100 all valid moves masked: uniform over valids */
            for (int a = 0; a < A; a++) nd->p[a] = nd->p[a] + (double)nd->valids[a];
            double s2 = pairwise_sum(nd->p, A);
            for (int a = 0; a < A; a++) nd->p[a] /= s2;
        }
        nd->ns = 0;
        nd->expanded = 1;
        m->stats[1]++;
        m->stats[5] += nv;
        val_t t = {(double)v32, ORC_F32};
        return t; /* :104 */
    }
    /* :106-121 PUCT over valid actions; strict '>' keeps the lowest index among maxima */
    double cur_best = -INFINITY;
    int best_act = -1, nvalid = 0;
    for (int a = 0; a < A; a++) {
        if (!nd->valids[a]) continue;
        nvalid++;
        double u;
        if (nd->nsa[a] > 0) /* (s,a) in Qsa */
            u = nd->q[a] + m->cpuct * nd->p[a] * sqrt((double)nd->ns) / (double)(1 + nd->nsa[a]);
        else
            u = m->cpuct * nd->p[a] * sqrt((double)nd->ns + 1e-8);
        if (u > cur_best) {
            cur_best = u;
            best_act = a;
        }
    }
    m->stats[3]++;
    m->stats[4] += nvalid;
    int a = best_act;
    uint8_t *next = (uint8_t *)malloc((size_t)m->keylen); /* :125-126 */
    memcpy(next, key, (size_t)m->keylen);
    orc_next_state(W, H, N, next, m->item_w, m->item_h, next + W * H, a);
    val_t v = search(m, next, nd->nsa[a] == 0 ? 1 : 0); /* :128; flag = first traversal of this edge (stats only) */
    free(next);
    int qk = nd->q_kind[a];
    orc_q_update(&nd->q[a], &qk, nd->nsa[a], v.v, v.kind); /* :130-136 */
    nd->q_kind[a] = (uint8_t)qk;
    nd->nsa[a] += 1;
    nd->ns += 1; /* :138 */
    return v;    /* :139 unchanged */
}

static void make_key(const orc_mcts *m, const uint8_t *board, const uint8_t *remaining, uint8_t *key) {
    memcpy(key, board, (size_t)(m->W * m->H));
    for (int i = 0; i < m->N; i++) key[m->W * m->H + i] = remaining[i] ? 1 : 0;
}

int orc_mcts_action_counts(orc_mcts *m, const uint8_t *board, const uint8_t *remaining, int n_sims, uint32_t *counts) {
    uint8_t *key = (uint8_t *)malloc((size_t)m->keylen);
    make_key(m, board, remaining, key);
    for (int i = 0; i < n_sims; i++) { /* :37-38 */
        m->stats[0]++;
        search(m, key, 0);
    }
    node *nd = lookup(m, key, 0, NULL);
    for (int a = 0; a < m->A; a++) counts[a] = (nd && nd->nsa) ? nd->nsa[a] : 0; /* :40-41 */
    free(key);
    return 0;
}

void orc_eval_uniform(void *user, const uint8_t *board, const uint8_t *remaining, float *pi, float *v) {
    (void)board; (void)remaining;
    int A = *(int *)user;
    float p = 1.0f / (float)A;
    for (int a = 0; a < A; a++) pi[a] = p;
    *v = 0.f;
}

int64_t orc_mcts_num_nodes(const orc_mcts *m) { return m->n_nodes; }

int orc_mcts_get_node(const orc_mcts *m, int64_t i, uint8_t *board, uint8_t *remaining, int *es, int *es_kind,
                      int *expanded, uint32_t *ns, uint8_t *valids, double *p, uint32_t *nsa, double *q,
                      uint8_t *q_kind) {
    if (i < 0 || i >= m->n_nodes) return -1;
    const node *nd = m->nodes[i];
    memcpy(board, nd->key, (size_t)(m->W * m->H));
    memcpy(remaining, nd->key + m->W * m->H, (size_t)m->N);
    *es = nd->has_es ? nd->es : 0;
    *es_kind = nd->es_kind;
    *expanded = nd->expanded;
    *ns = nd->ns;
    if (nd->expanded) {
        memcpy(valids, nd->valids, (size_t)m->A);
        memcpy(p, nd->p, sizeof(double) * (size_t)m->A);
        memcpy(nsa, nd->nsa, sizeof(uint32_t) * (size_t)m->A);
        memcpy(q, nd->q, sizeof(double) * (size_t)m->A);
        memcpy(q_kind, nd->q_kind, (size_t)m->A);
    }
    return 0;
}

void orc_mcts_stats(const orc_mcts *m, int64_t *out8) {
    memcpy(out8, m->stats, sizeof(m->stats));
    out8[7] = m->n_nodes;
}

/* ------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
uint64_t orc_sample_u64(uint64_t seed, uint64_t episode_id, uint64_t move) {
    return splitmix64(splitmix64(splitmix64(seed) ^ episode_id) ^ move);
}

/* CoachBPP.executeEpisode (CoachBPP.py:50-99) with a deterministic move rule in
 * place of np.random.seed() + np.random.choice (:86-87). */
int orc_play_episode(orc_mcts *m, int n_sims, int policy, uint64_t seed, uint64_t episode_id, int32_t *actions_out,
                     uint32_t *counts_out, int *outcome, double *score) {
    const int W = m->W, H = m->H, N = m->N, A = m->A;
    uint8_t *board = (uint8_t *)calloc((size_t)(W * H), 1); /* getInitBoard (:67) */
    uint8_t *remaining = (uint8_t *)malloc((size_t)N);
    uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)A);
    memset(remaining, 1, (size_t)N); /* getInitItems (:68) */
    int moves = 0;
    for (;;) {
        orc_mcts_action_counts(m, board, remaining, n_sims, counts); /* :76/:78 */
        if (counts_out) memcpy(counts_out + (size_t)moves * A, counts, sizeof(uint32_t) * (size_t)A);
        int action = -1;
        if (policy == 0) {
            uint32_t best = 0;
            for (int a = 0; a < A; a++)
                if (counts[a] > best) { best = counts[a]; action = a; }
        } else {
            uint64_t total = 0;
            for (int a = 0; a < A; a++) total += counts[a];
            uint64_t x = orc_sample_u64(seed, episode_id, (uint64_t)moves);
            uint64_t r = (uint64_t)(((__uint128_t)x * total) >> 64), acc = 0;
            for (int a = 0; a < A; a++) {
                acc += counts[a];
                if (acc > r) { action = a; break; }
            }
        }
        if (action < 0) break; /* no visit counts: root was terminal */
        actions_out[moves++] = action;
        orc_next_state(W, H, N, board, m->item_w, m->item_h, remaining, action); /* :88 */
        double r;
        int e = orc_game_ended(W, H, N, board, m->item_w, m->item_h, remaining, m->total_area, m->max_h, m->rewards,
                               m->n_rewards, m->alpha, &r); /* :91 */
        if (e != 0) {
            if (e == ORC_TIE) e = m->tie ? m->tie(m->tie_user, board, remaining) : 1;
            *outcome = e;
            *score = r;
            break;
        }
    }
    free(board); free(remaining); free(counts);
    return moves;
}
