/* Vectorised PtG environment step for the lock-step tree search (include/mctsq_tree.h).
 * Restates PTGEnv.step / reset of the `_per` environment (/root/reference/env/ptg_gym_env_per.py:102-195, 281-296,
 * 342-420, 422-662, 707-992) over state vectors: the mode table of :487-588 as a 5 x 5 dispatch, the five mode
 * changes, the window of one simulation step over a data set and its successor (:727-773) as an index range instead
 * of a copied array, the reward and the normalised observation.  Host code (gcc), built into libmctsq_tree.so. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/mctsq_tree.h"

enum { STARTUP_COLD, STARTUP_HOT, COOLDOWN, STANDBY_DOWN, STANDBY_UP, OP1, OP2, OP3, OP4, OP5, OP6, OP7, OP8, OP9,
       OP10, OP11, OP12, NDS };
enum { S_STANDBY, S_COOLDOWN, S_STARTUP, S_PARTIAL, S_FULL };
#define NOW 12                  /* position of the current hour in the price window (:316, 337, 384) */
#define HEAD MCTSQ_PTG_STATE_HEAD

typedef struct {
    mctsq_ptg_params p;
    int64_t D, off[NDS], len[NDS], U[NDS];
    double* op;                 /* [R][7] */
    double* opn;                /* [R][6]: columns 1..6 of every row, normalised (norm_col) once at creation: the look-back
                                   window is then ten 48-byte copies instead of sixty divisions per step */
    long double* cs;            /* prefix sums of columns 2..6, one leading zero row per data set: [R + NDS][5] */
    double* uv[NDS];            /* distinct T_cat values of a data set, ascending */
    int64_t* umin[NDS];         /* first row holding that value */
    double *el_n, *el_act, *pot, *pf, *g0, *g1;
} Env;

static inline double norm_col(const mctsq_ptg_params* p, int col, double x) {     /* col 1..6 of a process row */
    return (x - p->lb[2 + col]) / (p->ub[2 + col] - p->lb[2 + col]);
}

/* ---- _get_index (:715-725): first row minimising |T_cat[row] - t| ------------------------------------------- */
typedef struct { double v; int64_t i; } VI;
static int cmp_vi(const void* a, const void* b) {
    const VI *x = (const VI*)a, *y = (const VI*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return x->i < y->i ? -1 : x->i > y->i;
}

static int build_index(Env* e, int d) {
    const int64_t n = e->len[d];
    VI* t = (VI*)malloc(sizeof(VI) * (size_t)n);
    if (!t) return 0;
    for (int64_t r = 0; r < n; ++r) { t[r].v = e->op[(e->off[d] + r) * 7 + 1]; t[r].i = r; }
    qsort(t, (size_t)n, sizeof(VI), cmp_vi);
    e->uv[d] = (double*)malloc(sizeof(double) * (size_t)n);
    e->umin[d] = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    if (!e->uv[d] || !e->umin[d]) { free(t); return 0; }
    int64_t u = 0;
    for (int64_t r = 0; r < n; ++r)
        if (r == 0 || t[r].v != t[r - 1].v) { e->uv[d][u] = t[r].v; e->umin[d][u] = t[r].i; ++u; }
    e->U[d] = u;
    free(t);
    return 1;
}

static int64_t nearest_row(const Env* e, int d, double t) {
    const double* uv = e->uv[d];
    const int64_t* um = e->umin[d];
    int64_t lo = 0, hi = e->U[d];
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (uv[mid] < t) lo = mid + 1; else hi = mid;
    }
    /* the distance grows monotonically away from t on both sides (also after rounding), so walk outwards while it
       does not exceed the best one: rows tied after rounding are all seen and the first of them wins, as argmin does */
    double bd = INFINITY;
    int64_t best = -1;
    for (int64_t q = lo - 1; q >= 0; --q) {
        const double dd = fabs(uv[q] - t);
        if (dd < bd) { bd = dd; best = um[q]; }
        else if (dd == bd) { if (um[q] < best) best = um[q]; }
        else break;
    }
    for (int64_t q = lo; q < e->U[d]; ++q) {
        const double dd = fabs(uv[q] - t);
        if (dd < bd) { bd = dd; best = um[q]; }
        else if (dd == bd) { if (um[q] < best) best = um[q]; }
        else break;
    }
    return best < 0 ? 0 : best;
}

/* ---- construction ------------------------------------------------------------------------------------------- */
void mctsq_ptg_destroy(void* env) {
    Env* e = (Env*)env;
    if (!e) return;
    for (int d = 0; d < NDS; ++d) { free(e->uv[d]); free(e->umin[d]); }
    free(e->op); free(e->opn); free(e->cs); free(e->el_n); free(e->el_act); free(e->pot); free(e->pf); free(e->g0); free(e->g1);
    free(e);
}

void* mctsq_ptg_create(const mctsq_ptg_params* p, const double* op, const int64_t* ds_len, const double* e_r_b,
                       const double* g_e) {
    if (!p || !op || !ds_len || !e_r_b || !g_e) return NULL;
    if (p->step_size < 1 || p->seq_length < 1 || p->seq_step < 1 || p->sim_step < 1 || p->price_len <= NOW ||
        p->H < 1 || p->Dd < 1 || (p->seq_length - 1) * p->seq_step > p->step_size - 1 || 43200 % p->sim_step != 0)
        return NULL;
    Env* e = (Env*)calloc(1, sizeof(Env));
    if (!e) return NULL;
    e->p = *p;
    e->D = HEAD + 6 * p->seq_length;
    int64_t R = 0;
    for (int d = 0; d < NDS; ++d) {
        if (ds_len[d] < p->step_size) { free(e); return NULL; }
        e->off[d] = R; e->len[d] = ds_len[d]; R += ds_len[d];
    }
    const int64_t H = p->H, Dd = p->Dd, P = p->price_len;
    e->op = (double*)malloc(sizeof(double) * (size_t)R * 7);
    e->cs = (long double*)malloc(sizeof(long double) * (size_t)(R + NDS) * 5);
    e->opn = (double*)malloc(sizeof(double) * (size_t)R * 6);
    e->el_n = (double*)malloc(sizeof(double) * (size_t)(H * P));
    e->el_act = (double*)malloc(sizeof(double) * (size_t)H);
    e->pot = (double*)malloc(sizeof(double) * (size_t)H);
    e->pf = (double*)malloc(sizeof(double) * (size_t)H);
    e->g0 = (double*)malloc(sizeof(double) * (size_t)Dd * 2);
    e->g1 = (double*)malloc(sizeof(double) * (size_t)Dd * 2);
    if (!e->op || !e->opn || !e->cs || !e->el_n || !e->el_act || !e->pot || !e->pf || !e->g0 || !e->g1) { mctsq_ptg_destroy(e); return NULL; }
    memcpy(e->op, op, sizeof(double) * (size_t)R * 7);
    for (int64_t r = 0; r < R; ++r)
        for (int c = 1; c <= 6; ++c) e->opn[r * 6 + c - 1] = norm_col(p, c, e->op[r * 7 + c]);
    for (int d = 0; d < NDS; ++d) {
        long double* c = e->cs + (e->off[d] + d) * 5;
        for (int q = 0; q < 5; ++q) c[q] = 0.0L;
        for (int64_t r = 0; r < e->len[d]; ++r)
            for (int q = 0; q < 5; ++q) c[(r + 1) * 5 + q] = c[r * 5 + q] + (long double)e->op[(e->off[d] + r) * 7 + 2 + q];
        if (!build_index(e, d)) { mctsq_ptg_destroy(e); return NULL; }
    }
    /* e_r_b [3][P][H] -> normalised price window per hour (:284), current price, potential reward, load flag */
    for (int64_t h = 0; h < H; ++h) {
        for (int64_t q = 0; q < P; ++q) e->el_n[h * P + q] = (e_r_b[(0 * P + q) * H + h] - p->lb[0]) / (p->ub[0] - p->lb[0]);
        e->el_act[h] = e_r_b[(0 * P + NOW) * H + h];
        e->pot[h] = e_r_b[(1 * P + NOW) * H + h];
        e->pf[h] = e_r_b[(2 * P + NOW) * H + h];
    }
    /* g_e [2][2][Dd]: [gas|eua][today|tomorrow][day] */
    for (int c = 0; c < 2; ++c)
        for (int64_t d = 0; d < Dd; ++d) {
            e->g0[c * Dd + d] = g_e[(c * 2 + 0) * Dd + d];
            e->g1[c * Dd + d] = g_e[(c * 2 + 1) * Dd + d];
        }
    return e;
}

/* ---- the clock-dependent part of an observation (:590-603, 284-286) ------------------------------------------ */
typedef struct { int64_t hidx, didx; int bad; double gas[3], eua[3]; } Clock;

static Clock clock_of(const Env* e, int64_t k, int64_t d0) {
    const mctsq_ptg_params* p = &e->p;
    const int64_t Dd = p->Dd, secs = k * p->sim_step;
    Clock c;
    c.bad = 0;
    c.hidx = 24 * d0 + secs / 3600;
    c.didx = d0 + secs / 86400;
    int64_t dprev, dnext_today = 1;
    if (secs < 43200) { dprev = d0 - 1; if (dprev < 0) dprev += Dd; }          /* :127-129, negative index wraps */
    else { dprev = d0 + (secs - 43200) / 86400; dnext_today = 0; }                /* last noon (:599-603) */
    if (c.hidx < 0 || c.hidx >= p->H || c.didx < 0 || c.didx >= Dd || dprev < 0 || dprev >= Dd || d0 < 0 || d0 >= Dd) {
        c.bad = 1;
        c.hidx = c.hidx < 0 ? 0 : (c.hidx >= p->H ? p->H - 1 : c.hidx);
        c.didx = c.didx < 0 ? 0 : (c.didx >= Dd ? Dd - 1 : c.didx);
        dprev = dprev < 0 ? 0 : (dprev >= Dd ? Dd - 1 : dprev);
        d0 = d0 < 0 ? 0 : (d0 >= Dd ? Dd - 1 : d0);
    }
    const double *gas0 = e->g0, *eua0 = e->g0 + Dd, *gas1 = e->g1, *eua1 = e->g1 + Dd;
    c.gas[0] = gas0[dprev]; c.eua[0] = eua0[dprev];
    c.gas[1] = gas0[c.didx]; c.eua[1] = eua0[c.didx];
    if (dnext_today) { c.gas[2] = gas0[d0]; c.eua[2] = eua0[d0]; }
    else { c.gas[2] = gas1[dprev]; c.eua[2] = eua1[dprev]; }
    return c;
}

static void write_rows(const Env* e, const Clock* c, const double* seq, float* price, float* proc, float* gas) {
    const mctsq_ptg_params* p = &e->p;
    const double* el = e->el_n + c->hidx * p->price_len;
    for (int64_t q = 0; q < p->price_len; ++q) price[q] = (float)el[q];
    for (int64_t q = 0; q < 6 * p->seq_length; ++q) proc[q] = (float)seq[q];
    for (int t = 0; t < 3; ++t) {
        gas[2 * t + 0] = (float)((c->gas[t] - p->lb[1]) / (p->ub[1] - p->lb[1]));
        gas[2 * t + 1] = (float)((c->eua[t] - p->lb[2]) / (p->ub[2] - p->lb[2]));
    }
}


/* ---- reset (:664-705 with :102-195) ------------------------------------------------------------------------- */
void mctsq_ptg_reset(const void* env, int64_t n, const int64_t* d0, double* states) {
    const Env* e = (const Env*)env;
    const int64_t D = e->D;
    const int64_t i0 = nearest_row(e, COOLDOWN, 16.0);
    const double* row = e->op + (e->off[COOLDOWN] + i0) * 7;
    for (int64_t s = 0; s < n; ++s) {
        double* st = states + s * D;
        st[0] = (double)i0; st[1] = 0; st[2] = 0; st[3] = S_COOLDOWN; st[4] = 16.0;
        st[5] = STANDBY_DOWN; st[6] = STARTUP_COLD; st[7] = OP1; st[8] = OP2; st[9] = 0;
        st[10] = (double)d0[s]; st[11] = 0.0;
        for (int64_t m = 0; m < e->p.seq_length; ++m)
            for (int c = 1; c <= 6; ++c) st[HEAD + m * 6 + c - 1] = norm_col(&e->p, c, row[c]);
    }
}

int64_t mctsq_ptg_observe(const void* env, int64_t n, const double* states, float* price, int64_t ld_price,
                          float* proc, int64_t ld_proc, float* gas, int64_t ld_gas, int64_t* meth) {
    const Env* e = (const Env*)env;
    int64_t bad = 0;
    for (int64_t s = 0; s < n; ++s) {
        const double* st = states + s * e->D;
        const Clock c = clock_of(e, (int64_t)st[2], (int64_t)st[10]);
        bad += c.bad;
        write_rows(e, &c, st + HEAD, price + s * ld_price, proc + s * ld_proc, gas + s * ld_gas);
        if (meth) meth[s] = (int64_t)st[3];
    }
    return bad;
}

int64_t mctsq_ptg_info(const void* env, int64_t n, const double* states, double* out) {
    const Env* e = (const Env*)env;
    int64_t bad = 0;
    for (int64_t s = 0; s < n; ++s) {
        const double* st = states + s * e->D;
        const Clock c = clock_of(e, (int64_t)st[2], (int64_t)st[10]);
        bad += c.bad;
        double* o = out + s * 8;
        o[0] = e->el_act[c.hidx]; o[1] = c.gas[1]; o[2] = c.eua[1]; o[3] = e->pot[c.hidx]; o[4] = e->pf[c.hidx];
        o[5] = st[11]; o[6] = st[4]; o[7] = st[9];
    }
    return bad;
}

/* ---- one step ----------------------------------------------------------------------------------------------- */
/* [action][state] -> 0: stay on the current data set (_cont), 1..5: change into standby / cooldown / startup /
 * partial load / full load (:487-588) */
static const int8_t DISPATCH[5][5] = {
    {0, 1, 1, 1, 1},
    {2, 0, 2, 2, 2},
    {3, 3, 0, 0, 0},
    {0, 0, 0, 0, 4},
    {0, 0, 0, 5, 0},
};

static inline int64_t noisy_row(int64_t idx, double nz) {         /* int(max(idx + N(0, noise), 0)) (:811, 825, 849) */
    double v = (double)idx + nz;
    if (!(v > 0.0)) v = 0.0;
    return (int64_t)v;
}

static int valid_ids(int m, int sb, int su, int pt, int fl) {
    const int pt_ok = pt == OP1 || pt == OP8 || pt == OP4 || pt == OP5 || pt == OP6 || pt == OP7;
    const int fl_ok = fl == OP2 || fl == OP3 || fl == OP9 || fl == OP10 || fl == OP11 || fl == OP12;
    return m >= 0 && m <= 4 && (sb == STANDBY_DOWN || sb == STANDBY_UP) && (su == STARTUP_COLD || su == STARTUP_HOT) &&
           pt_ok && fl_ok;
}

static int step_one(const Env* e, const double* st, int64_t action, double nz, double* nx, float* price, float* proc,
                    float* gas, double* reward, uint8_t* term, int64_t* meth) {
    const mctsq_ptg_params* p = &e->p;
    const int64_t step = p->step_size;
    int64_t i = (int64_t)st[0], j = (int64_t)st[1];
    const int64_t k = (int64_t)st[2], d0 = (int64_t)st[10];
    int m = (int)st[3], sb = (int)st[5], su = (int)st[6], pt = (int)st[7], fl = (int)st[8], hot = (int)st[9];
    double t_cat = st[4];
    if (action < 0 || action > 4 || !valid_ids(m, sb, su, pt, fl) || i < 0 || j < 0) {
        memcpy(nx, st, sizeof(double) * (size_t)e->D);
        *reward = 0.0; *term = 1; if (meth) *meth = m;
        return 1;
    }
    if (t_cat <= p->t_cat_startup_cold) hot = 0;                  /* :454-457 */
    else if (t_cat >= p->t_cat_startup_hot) hot = 1;

    int a, b, next_state, change = 0;
    switch (DISPATCH[action][m]) {
    case 0:                                                       /* _cont (:775-795) */
        j += 1;
        next_state = m;
        if (m == S_STANDBY) a = b = sb;
        else if (m == S_COOLDOWN) a = b = COOLDOWN;
        else if (m == S_STARTUP) { a = su; b = pt; next_state = S_PARTIAL; change = 1; }
        else if (m == S_PARTIAL) a = b = pt;
        else a = b = fl;
        break;
    case 1:                                                       /* _standby (:797-816) */
        m = next_state = S_STANDBY;
        sb = t_cat <= p->t_cat_standby ? STANDBY_UP : STANDBY_DOWN;
        i = noisy_row(nearest_row(e, sb, t_cat), nz); j = 1;
        a = b = sb;
        break;
    case 2:                                                       /* _cooldown (:818-830) */
        m = next_state = S_COOLDOWN;
        i = noisy_row(nearest_row(e, COOLDOWN, t_cat), nz); j = 1;
        a = b = COOLDOWN;
        break;
    case 3:                                                       /* _startup (:832-854) */
        m = S_STARTUP; next_state = S_PARTIAL; change = 1;
        pt = OP1; fl = OP2;
        su = hot == 0 ? STARTUP_COLD : STARTUP_HOT;
        i = noisy_row(nearest_row(e, su, t_cat), nz); j = 1;
        a = su; b = pt;
        break;
    case 4: {                                                     /* _partial (:856-923), coming from full load */
        m = next_state = S_PARTIAL;
        const double t = (double)(i + j * step);
        if (fl == OP2) {
            if (t < p->time2_start_f_p) { pt = OP1; i = nearest_row(e, OP1, t_cat); j = 1; }
            else { pt = OP8; i = 0; j = 1; }
        } else if (fl == OP3) {
            if (t < p->time1_p_f_p) { pt = OP8; i = (int64_t)p->i_fully_developed; j = (int64_t)p->j_fully_developed; }
            else if (p->time1_p_f_p < t && t < p->time2_p_f_p) { pt = OP4; j += 1; }
            else if (p->time2_p_f_p < t && t < p->time_p_f) { pt = OP4; i = (int64_t)p->time2_p_f_p; j = 1; }
            else if (p->time_p_f < t && t < p->time34_p_f_p) { pt = OP5; i = (int64_t)p->time3_p_f_p; j = 1; }
            else if (p->time34_p_f_p < t && t < p->time45_p_f_p) { pt = OP6; i = (int64_t)p->time4_p_f_p; j = 1; }
            else if (p->time45_p_f_p < t && t < p->time5_p_f_p) { pt = OP7; i = (int64_t)p->time5_p_f_p; j = 1; }
            else { pt = OP8; i = 0; j = 1; }
        } else { pt = OP8; i = 0; j = 1; }
        a = b = pt;
        break;
    }
    default: {                                                    /* _full (:925-992), coming from partial load */
        m = next_state = S_FULL;
        const double t = (double)(i + j * step);
        if (pt == OP1) { fl = t < p->time1_start_p_f ? OP2 : OP3; i = 0; j = 1; }
        else if (pt == OP8) {
            if (t < p->time1_f_p_f) { fl = OP3; i = (int64_t)p->i_fully_developed; j = (int64_t)p->j_fully_developed; }
            else if (p->time1_f_p_f < t && t < p->time_f_p) { fl = OP9; j += 1; }
            else if (p->time_f_p < t && t < p->time23_f_p_f) { fl = OP9; i = (int64_t)p->time2_f_p_f; j = 1; }
            else if (p->time23_f_p_f < t && t < p->time34_f_p_f) { fl = OP10; i = (int64_t)p->time3_f_p_f; j = 1; }
            else if (p->time34_f_p_f < t && t < p->time45_f_p_f) { fl = OP11; i = (int64_t)p->time4_f_p_f; j = 1; }
            else if (p->time45_f_p_f < t && t < p->time5_f_p_f) { fl = OP12; i = (int64_t)p->time5_f_p_f; j = 1; }
            else { fl = OP3; i = 0; j = 1; }
        } else { fl = OP3; i = 0; j = 1; }
        a = b = fl;
        break;
    }
    }

    /* _perform_sim_step (:727-773): the step's rows are rows [i_start, i_start + n1) of data set a followed by
       step - n1 rows that are either the first rows of data set b or copies of a's last row */
    const int64_t total = e->len[a], end = i + j * step, i_start = i + (j - 1) * step;
    int64_t n1 = step;
    int tail_b = 0;
    if (end >= total) {
        m = next_state;
        const int64_t over = end - total;
        if (over < step) {
            n1 = step - over;
            if (change) { tail_b = 1; i = over; j = 0; }
        } else n1 = 0;
    }
    const double* A = e->op + e->off[a] * 7;
    const double* B = e->op + e->off[b] * 7;
    const double* last = A + (total - 1) * 7;
    const int64_t n2 = step - n1;

    long double sum[5] = {0, 0, 0, 0, 0};
    if (n1 > 0) {
        const long double* c = e->cs + (e->off[a] + a) * 5;
        for (int q = 0; q < 5; ++q) sum[q] = c[(i_start + n1) * 5 + q] - c[i_start * 5 + q];
    }
    if (n2 > 0) {
        if (tail_b) {
            const long double* c = e->cs + (e->off[b] + b) * 5;
            for (int q = 0; q < 5; ++q) sum[q] += c[n2 * 5 + q];
        } else
            for (int q = 0; q < 5; ++q) sum[q] += (long double)n2 * (long double)last[2 + q];
    }
    double avg[5];
    for (int q = 0; q < 5; ++q) avg[q] = (double)sum[q] / (double)step;         /* np.average (:611-615) */

    /* strided look-back window (:619-628), newest first */
    {
        const double* An = e->opn + e->off[a] * 6;
        const double* Bn = e->opn + e->off[b] * 6;
        const double* lastn = An + (total - 1) * 6;
        for (int64_t w = 0; w < p->seq_length; ++w) {
            const int64_t pos = step - 1 - w * p->seq_step;
            const double* row = pos < n1 ? An + (i_start + pos) * 6 : (tail_b ? Bn + (pos - n1) * 6 : lastn);
            memcpy(nx + HEAD + w * 6, row, sizeof(double) * 6);
        }
    }
    const double* lastrow = n2 > 0 ? (tail_b ? B + (n2 - 1) * 7 : last) : A + (i_start + step - 1) * 7;
    t_cat = lastrow[1];                                                          /* :609 */

    const Clock c = clock_of(e, k + 1, d0);
    const double el = e->el_act[c.hidx], gas_now = c.gas[1], eua_now = c.eua[1];
    const double h2 = avg[0], ch4 = avg[1], h2_res = avg[2], h2o = avg[3], heating = avg[4];

    /* _get_reward (:342-420), same operation order */
    const double b_s3 = p->scenario3 ? 1.0 : 0.0;
    const double ch4_vol = ch4 * p->convert_mol_to_Nm3;
    const double h2_res_vol = h2_res * p->convert_mol_to_Nm3;
    const double q_ch4 = ch4_vol * p->H_u_CH4 * 1000;
    const double q_h2_res = h2_res_vol * p->H_u_H2 * 1000;
    const double ch4_rev = (q_ch4 + q_h2_res) * gas_now;
    const double power_chp = q_ch4 * p->eta_CHP * b_s3;
    const double q_chp = q_ch4 * (1 - p->eta_CHP) * b_s3;
    const double chp_rev = power_chp * p->eeg_el_price;
    const double q_steam = h2o * (p->dt_water * p->cp_water + p->h_H2O_evap) / 3600;
    const double steam_rev = (q_steam + q_chp) * p->heat_price;
    const double h2_vol = h2 * p->convert_mol_to_Nm3;
    const double o2_vol = 1.0 / 2 * h2_vol * 3600;
    const double o2_rev = o2_vol * p->o2_price;
    const double co2 = ch4 * p->Molar_mass_CO2 / 1000;
    const double eua_rev = co2 / 1000 * 3600 * eua_now * 100;
    const double cost_heat = heating / 1000 * el;
    const double load = h2_vol / p->max_h2_volumeflow;
    double eta;
    if (load < p->min_load_electrolyzer) eta = 0.02;
    else
        eta = (0.598 - 0.325 * pow(load, 2) + 0.218 * pow(load, 3) + 0.01 * pow(load, -1) -
               1.68 * 0.001 * pow(load, -2) + 2.51 * 1e-05 * pow(load, -3));
    const double cost_el = h2_vol * p->H_u_H2 * 1000 / eta * el;
    const double elec_costs = cost_heat + cost_el;
    const double water_elec = h2 * p->Molar_mass_H2O / 1000 * 3600;
    const double water_costs = (h2o + water_elec) / p->rho_water * p->water_price;
    const double rew = (ch4_rev + chp_rev + steam_rev + eua_rev + o2_rev - elec_costs - water_costs) *
                       (double)p->sim_step / 3600;
    *reward = (rew - p->rew_l_b) / (p->rew_u_b - p->rew_l_b);
    *term = (uint8_t)(k == p->eps_sim_steps - 6);                                /* :707-712, before k += 1 */
    if (meth) *meth = m;

    nx[0] = (double)i; nx[1] = (double)j; nx[2] = (double)(k + 1); nx[3] = m; nx[4] = t_cat;
    nx[5] = sb; nx[6] = su; nx[7] = pt; nx[8] = fl; nx[9] = hot; nx[10] = (double)d0; nx[11] = st[11] + rew;
    if (price) write_rows(e, &c, nx + HEAD, price, proc, gas);
    return c.bad;
}

int64_t mctsq_ptg_step(const void* env, int64_t n, const double* states, const int64_t* actions, const double* noise,
                       double* next, float* price, int64_t ld_price, float* proc, int64_t ld_proc, float* gas,
                       int64_t ld_gas, double* reward, uint8_t* term, int64_t* meth) {
    const Env* e = (const Env*)env;
    const int64_t D = e->D;
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) if (n >= 128) num_threads(mctsq_tree_threads())
    for (int64_t s = 0; s < n; ++s)
        bad += step_one(e, states + s * D, actions[s], noise ? noise[s] : 0.0, next + s * D,
                        price ? price + s * ld_price : NULL, proc ? proc + s * ld_proc : NULL,
                        gas ? gas + s * ld_gas : NULL, reward + s, term + s, meth ? meth + s : NULL);
    return bad;
}

/* used by lockstep.c: one state, caller-provided output slots */
int mctsq_ptg_step_one(const void* env, const double* st, int64_t action, double nz, double* nx, float* price,
                       float* proc, float* gas, double* reward, uint8_t* term, int64_t* meth) {
    return step_one((const Env*)env, st, action, nz, nx, price, proc, gas, reward, term, meth);
}

int64_t mctsq_ptg_state_dim(const void* env) { return ((const Env*)env)->D; }
