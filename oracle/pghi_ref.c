/*
 * oracle/pghi_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's phase-gradient heap integration
 * (PGHI), offline and realtime, with the reference's exact binary-heap order.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (acids_transforms_amd) never does.
 *
 * What it follows (all paths relative to /root/reference/acids_transforms):
 *   heap            utils/heapq.py:9-59   (CPython heapq semantics, keys only,
 *                                          strict '<', right child on ties)
 *   offline  grad   transforms/dgt.py:222-236  (DGT.modgabphasegrad)
 *   offline  hgi    transforms/dgt.py:156-162, 168-220 (DGT.pghi/perform_hgi)
 *   realtime grad   transforms/dgt.py:378-397  (RealtimeDGT.modgabphasegrad)
 *   realtime hgi    transforms/dgt.py:338-354, 399-466
 *
 * Parity is pinned by tests/golden/g4_pghi_offline.npz, g5_rtpghi_kernel.npz
 * (outputs of the reference itself run in the build container, including the
 * recorded heap pop order) -- see tests/test_oracle_golden.py.
 *
 * Arithmetic is fp32 with the reference's operation order; compile with
 * -ffp-contract=off so no FMA contraction changes a rounding.
 */
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float key;   /* -magnitude */
    int32_t idx; /* row * F + col */
} item_t;

/* utils/heapq.py:9-21 */
static void siftdown(item_t *h, int startpos, int pos)
{
    item_t newitem = h[pos];
    while (pos > startpos) {
        int parentpos = (pos - 1) >> 1;
        item_t parent = h[parentpos];
        if (newitem.key < parent.key) {
            h[pos] = parent;
            pos = parentpos;
            continue;
        }
        break;
    }
    h[pos] = newitem;
}

/* utils/heapq.py:24-42 */
static void siftup(item_t *h, int endpos, int pos)
{
    int startpos = pos;
    item_t newitem = h[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        int rightpos = childpos + 1;
        if (rightpos < endpos && !(h[childpos].key < h[rightpos].key))
            childpos = rightpos;
        h[pos] = h[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    h[pos] = newitem;
    siftdown(h, startpos, pos);
}

/* utils/heapq.py:45-48 */
static void heappush(item_t *h, int *n, float key, int32_t idx)
{
    h[*n].key = key;
    h[*n].idx = idx;
    (*n)++;
    siftdown(h, 0, *n - 1);
}

/* utils/heapq.py:51-59 */
static item_t heappop(item_t *h, int *n)
{
    item_t last = h[*n - 1];
    (*n)--;
    if (*n > 0) {
        item_t ret = h[0];
        h[0] = last;
        siftup(h, *n, 0);
        return ret;
    }
    return last;
}

/* ---------------------------------------------------------------------- */
/* offline gradients: dgt.py:222-236.  s = clamp(mag, eps) is computed by
 * the caller (dgt.py:157).  Y is log(s) replicate-padded by one on all four
 * sides; dxdw is the central difference along the LAST axis (frequency),
 * dxdt along axis 0 (time).                                               */
void pghi_offline_grad_ref(const float *s, int T, int F, float gamma, int n_fft, int hop,
                           float *tgradw, float *fgradw)
{
    /* fmul = gamma / (hop * n_fft): fp32 / int64 product  (dgt.py:223) */
    const float fmul = gamma / (float)((int64_t)hop * (int64_t)n_fft);
    /* 2*pi*hop/n_fft: python float * int64 tensor -> fp32, then / int64  (dgt.py:233) */
    const float two_pi_f = (float)(2.0 * 3.14159265358979323846);
    const float fstep = (two_pi_f * (float)hop) / (float)n_fft;
    const float pi_f = (float)3.14159265358979323846;
    float *Y = (float *)malloc(sizeof(float) * (size_t)T * F);
    for (long i = 0; i < (long)T * F; i++)
        Y[i] = logf(s[i]);
    for (int t = 0; t < T; t++) {
        const float *row = Y + (long)t * F;
        const float *up = Y + (long)(t + 1 < T ? t + 1 : T - 1) * F;
        const float *dn = Y + (long)(t > 0 ? t - 1 : 0) * F;
        for (int k = 0; k < F; k++) {
            float right = row[k + 1 < F ? k + 1 : F - 1];
            float left = row[k > 0 ? k - 1 : 0];
            float dxdw = (right - left) / 2.0f;
            float dxdt = (up[k] - dn[k]) / 2.0f;
            fgradw[(long)t * F + k] = dxdw / fmul + fstep * (float)k;
            tgradw[(long)t * F + k] = (-fmul) * dxdt + pi_f;
        }
    }
    free(Y);
}

/* dgt.py:168-220.  `spec` is the clamped magnitude (modified in place, as the
 * reference does on its clone).  order_out (optional) receives row*F+col of
 * every heappop in sequence.  Returns the number of pops.                   */
long pghi_offline_hgi_ref(float *spec, int T, int F, const float *tgradw, const float *fgradw,
                          float abstol, float tol, float *phase, int32_t *order_out)
{
    const long n = (long)T * F;
    long npops = 0;
    memset(phase, 0, sizeof(float) * n);
    if (n == 0)
        return 0;
    item_t *heap = (item_t *)malloc(sizeof(item_t) * (n + 2));
    int hn = 0;
    /* :173-176 first (row-major) argmax */
    float max_val = spec[0];
    long max_pos = 0;
    for (long i = 1; i < n; i++)
        if (spec[i] > max_val) {
            max_val = spec[i];
            max_pos = i;
        }
    heap[hn].key = -max_val;
    heap[hn].idx = (int32_t)max_pos;
    hn++;
    spec[max_pos] = abstol;
    /* :177-178 relative threshold */
    {
        const float thr = max_val * tol;
        for (long i = 0; i < n; i++)
            if (spec[i] < thr)
                spec[i] = abstol;
    }
    while (max_val > abstol) { /* :179 */
        while (hn > 0) {       /* :180 */
            item_t it = heappop(heap, &hn);
            if (order_out)
                order_out[npops] = it.idx;
            npops++;
            const int col = it.idx / F; /* frame index ("col" in the reference) */
            const int row = it.idx % F; /* bin index */
            const long c = it.idx;
            /* :188-194 next frame, fgradw */
            if (col < T - 1 && spec[c + F] > abstol) {
                phase[c + F] = phase[c] + (fgradw[c] + fgradw[c + F]) / 2.0f;
                heappush(heap, &hn, -spec[c + F], (int32_t)(c + F));
                spec[c + F] = abstol;
            }
            /* :195-201 previous frame */
            if (col > 0 && spec[c - F] > abstol) {
                phase[c - F] = phase[c] - (fgradw[c] + fgradw[c - F]) / 2.0f;
                heappush(heap, &hn, -spec[c - F], (int32_t)(c - F));
                spec[c - F] = abstol;
            }
            /* :202-208 next bin, tgradw */
            if (row < F - 1 && spec[c + 1] > abstol) {
                phase[c + 1] = phase[c] + (tgradw[c] + tgradw[c + 1]) / 2.0f;
                heappush(heap, &hn, -spec[c + 1], (int32_t)(c + 1));
                spec[c + 1] = abstol;
            }
            /* :209-215 previous bin */
            if (row > 0 && spec[c - 1] > abstol) {
                phase[c - 1] = phase[c] - (tgradw[c] + tgradw[c - 1]) / 2.0f;
                heappush(heap, &hn, -spec[c - 1], (int32_t)(c - 1));
                spec[c - 1] = abstol;
            }
        }
        /* :216-219 reseed from the global max of what is left */
        max_val = spec[0];
        max_pos = 0;
        for (long i = 1; i < n; i++)
            if (spec[i] > max_val) {
                max_val = spec[i];
                max_pos = i;
            }
        heappush(heap, &hn, -max_val, (int32_t)max_pos);
        spec[max_pos] = abstol;
    }
    free(heap);
    return npops;
}

/* DGT.pghi (dgt.py:156-162) for one (T,F) magnitude array.                 */
long pghi_offline_ref(const float *mag, int T, int F, float gamma, int n_fft, int hop, float tol,
                      float eps, float *phase, float *tgradw_out, float *fgradw_out,
                      int32_t *order_out)
{
    const long n = (long)T * F;
    float *s = (float *)malloc(sizeof(float) * (n ? n : 1));
    float *tg = tgradw_out ? tgradw_out : (float *)malloc(sizeof(float) * (n ? n : 1));
    float *fg = fgradw_out ? fgradw_out : (float *)malloc(sizeof(float) * (n ? n : 1));
    for (long i = 0; i < n; i++)
        s[i] = mag[i] < eps ? eps : mag[i]; /* :157 clamp */
    pghi_offline_grad_ref(s, T, F, gamma, n_fft, hop, tg, fg);
    long np = pghi_offline_hgi_ref(s, T, F, tg, fg, eps, tol, phase, order_out);
    free(s);
    if (!tgradw_out)
        free(tg);
    if (!fgradw_out)
        free(fg);
    return np;
}

/* ---------------------------------------------------------------------- */
/* realtime gradients: dgt.py:378-397 on the (R = n+2, F) clamped stack of
 * [2 history frames ; n new frames].  Time-border rows of the reference's
 * torch.empty array are DEFINED as 0 here (SURVEY.md hard part 3).         */
void pghi_rt_grad_ref(const float *s, int R, int F, float gamma, int n_fft, int hop, float *tgradw,
                      float *fgradw)
{
    const float fmul = gamma / (float)((int64_t)hop * (int64_t)n_fft);
    const float two_pi_f = (float)(2.0 * 3.14159265358979323846);
    const float fstep = (two_pi_f * (float)hop) / (float)n_fft;
    const float pi_f = (float)3.14159265358979323846;
    float *Y = (float *)malloc(sizeof(float) * (size_t)R * F);
    for (long i = 0; i < (long)R * F; i++)
        Y[i] = logf(s[i]);
    for (int j = 0; j < R; j++) {
        const float *row = Y + (long)j * F;
        for (int k = 0; k < F; k++) {
            float right = row[k + 1 < F ? k + 1 : F - 1];
            float left = row[k > 0 ? k - 1 : 0];
            float dxdw = (right - left) / 2.0f;                      /* :393 */
            float nxt = (j + 1 < R) ? Y[(long)(j + 1) * F + k] : 0.0f; /* Y[j+2] */
            float prv = (j > 0) ? Y[(long)(j - 1) * F + k] : 0.0f;     /* Y[j]   */
            float dxdt = (3.0f * nxt - 4.0f * row[k] + prv) / 2.0f;  /* :394 */
            fgradw[(long)j * F + k] = dxdw / fmul + fstep * (float)k; /* :395 */
            tgradw[(long)j * F + k] = (-fmul) * dxdt + pi_f;          /* :396 */
        }
    }
    free(Y);
}

/* dgt.py:399-466 for ONE stream.  spec: (R,F) clamped magnitudes (modified).
 * prev_phase: (F).  tgradw/fgradw: (R,F) as returned by pghi_rt_grad_ref; the
 * reference front-pads them with two zero rows (:408-410) so "row r" of the
 * padded arrays is row r-2 here.  noise: (R-2,F) standard normal draws used
 * for the bins at or below abstol (:404-405).  phase_out: (R-2,F).
 * order_out (optional, capacity >= 4*R*F) receives row*F+col per pop.       */
long pghi_rt_hgi_ref(float *spec, int R, int F, const float *prev_phase, const float *tgradw,
                     const float *fgradw, float tol, float eps, const float *noise,
                     float *phase_out, int32_t *order_out)
{
    const long n = (long)R * F;
    long npops = 0;
    /* :400 abstol = clamp(tol * max(spec), eps) */
    float smax = spec[0];
    for (long i = 1; i < n; i++)
        if (spec[i] > smax)
            smax = spec[i];
    float abstol = tol * smax;
    if (abstol < eps)
        abstol = eps;
    float *phase = (float *)calloc((size_t)n, sizeof(float));
    float *hist = (float *)malloc(sizeof(float) * n);
    memcpy(hist, spec, sizeof(float) * n); /* :411 */
    for (int k = 0; k < F; k++)
        phase[F + k] = prev_phase[k]; /* :403 */
    for (long i = 2L * F; i < n; i++)  /* :404-405 */
        phase[i] = (spec[i] > abstol) ? 0.0f : noise[i - 2L * F];
#define TG(r, k) ((r) >= 2 ? tgradw[(long)((r)-2) * F + (k)] : 0.0f)
#define FG(r, k) ((r) >= 2 ? fgradw[(long)((r)-2) * F + (k)] : 0.0f)
    item_t *heap = (item_t *)malloc(sizeof(item_t) * (4L * F + 8));
    for (int f = 2; f < R; f++) { /* :413 */
        float *row = spec + (long)f * F;
        float max_val = row[0];
        int max_k = 0;
        for (int k = 1; k < F; k++)
            if (row[k] > max_val) {
                max_val = row[k];
                max_k = k;
            }
        if (max_val <= abstol)
            continue; /* :416-417 */
        int hn = 0;
        heap[hn].key = -max_val; /* :427 seed is NOT marked visited */
        heap[hn].idx = f * F + max_k;
        hn++;
        for (int k = 0; k < F; k++) /* :428-430 */
            if (hist[(long)(f - 1) * F + k] > abstol)
                heappush(heap, &hn, -hist[(long)(f - 1) * F + k], (f - 1) * F + k);
        while (max_val > abstol) { /* :433 */
            while (hn > 0) {
                item_t it = heappop(heap, &hn);
                if (order_out)
                    order_out[npops] = it.idx;
                npops++;
                const int r = it.idx / F, k = it.idx % F;
                if (r == f - 1) { /* :436-443 propagate in time with tgradw */
                    if (row[k] > abstol) {
                        phase[(long)f * F + k] =
                            phase[(long)(f - 1) * F + k] + 0.5f * (TG(f - 1, k) + TG(f, k));
                        heappush(heap, &hn, -row[k], f * F + k);
                        row[k] = abstol;
                    }
                }
                if (r == f) { /* :444-460 propagate in frequency with fgradw */
                    if (k + 1 < F) {
                        if (row[k + 1] > abstol) {
                            phase[(long)f * F + k + 1] =
                                phase[(long)f * F + k] + 0.5f * (FG(f, k) + FG(f, k + 1));
                            heappush(heap, &hn, -row[k + 1], f * F + k + 1);
                            row[k + 1] = abstol;
                        }
                    }
                    if (k - 1 > 0) { /* bin 0 is never reached downward (:453) */
                        if (row[k - 1] > abstol) {
                            phase[(long)f * F + k - 1] =
                                phase[(long)f * F + k] - 0.5f * (FG(f, k) + FG(f, k - 1));
                            heappush(heap, &hn, -row[k - 1], f * F + k - 1);
                            row[k - 1] = abstol;
                        }
                    }
                }
            }
            /* :461-465 reseed inside the frame */
            max_val = row[0];
            max_k = 0;
            for (int k = 1; k < F; k++)
                if (row[k] > max_val) {
                    max_val = row[k];
                    max_k = k;
                }
            heappush(heap, &hn, -max_val, f * F + max_k);
            row[max_k] = abstol;
        }
    }
#undef TG
#undef FG
    memcpy(phase_out, phase + 2L * F, sizeof(float) * (n - 2L * F));
    free(heap);
    free(hist);
    free(phase);
    return npops;
}

/* RealtimeDGT.pghi (dgt.py:338-354) for S streams: mag_hist (S,2,F), mag (S,n,F),
 * prev_phase (S,F), noise (S,n,F) -> phase (S,n,F).                        */
void pghi_rt_ref(const float *mag_hist, const float *mag, const float *prev_phase,
                 const float *noise, int S, int n, int F, float gamma, int n_fft, int hop,
                 float tol, float eps, float *phase, float *tgradw_out, float *fgradw_out)
{
    const int R = n + 2;
    float *s = (float *)malloc(sizeof(float) * (size_t)R * F);
    float *tg = (float *)malloc(sizeof(float) * (size_t)R * F);
    float *fg = (float *)malloc(sizeof(float) * (size_t)R * F);
    for (int i = 0; i < S; i++) {
        for (long j = 0; j < 2L * F; j++) {
            float v = mag_hist[(long)i * 2 * F + j];
            s[j] = v < eps ? eps : v;
        }
        for (long j = 0; j < (long)n * F; j++) {
            float v = mag[(long)i * n * F + j];
            s[2L * F + j] = v < eps ? eps : v;
        }
        pghi_rt_grad_ref(s, R, F, gamma, n_fft, hop, tg, fg);
        if (tgradw_out)
            memcpy(tgradw_out + (long)i * R * F, tg, sizeof(float) * (size_t)R * F);
        if (fgradw_out)
            memcpy(fgradw_out + (long)i * R * F, fg, sizeof(float) * (size_t)R * F);
        pghi_rt_hgi_ref(s, R, F, prev_phase + (long)i * F, tg, fg, tol, eps,
                        noise + (long)i * n * F, phase + (long)i * n * F, NULL);
    }
    free(s);
    free(tg);
    free(fg);
}

/* batch driver used by bench.py's cpu_baseline leg: B clips, one after another */
long pghi_offline_batch_ref(const float *mag, int B, int T, int F, float gamma, int n_fft, int hop,
                            float tol, float eps, float *phase)
{
    long total = 0;
    for (int b = 0; b < B; b++)
        total += pghi_offline_ref(mag + (long)b * T * F, T, F, gamma, n_fft, hop, tol, eps,
                                  phase + (long)b * T * F, NULL, NULL, NULL);
    return total;
}

/* The same batch on `nthreads` host threads, one clip per thread at a time (clips are handed out by an
 * atomic counter): the all-cores CPU baseline of bench.py (SURVEY.md 8d "C++ PGHI ... one thread per clip").
 * The per-clip arithmetic is pghi_offline_ref itself, so the results are those of the serial driver. */
typedef struct {
    const float *mag;
    float *phase;
    int B, T, F, n_fft, hop;
    float gamma, tol, eps;
    atomic_int next;
    atomic_long total;
} mt_job_t;

static void *mt_worker(void *arg)
{
    mt_job_t *j = (mt_job_t *)arg;
    const long n = (long)j->T * j->F;
    for (;;) {
        int b = atomic_fetch_add(&j->next, 1);
        if (b >= j->B)
            break;
        long np = pghi_offline_ref(j->mag + b * n, j->T, j->F, j->gamma, j->n_fft, j->hop, j->tol, j->eps,
                                   j->phase + b * n, NULL, NULL, NULL);
        atomic_fetch_add(&j->total, np);
    }
    return NULL;
}

long pghi_offline_batch_mt_ref(const float *mag, int B, int T, int F, float gamma, int n_fft, int hop,
                               float tol, float eps, float *phase, int nthreads)
{
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > B)
        nthreads = B > 0 ? B : 1;
    mt_job_t job = {mag, phase, B, T, F, n_fft, hop, gamma, tol, eps, 0, 0};
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    int started = 0;
    for (int i = 0; i < nthreads; i++)
        if (pthread_create(&th[i], NULL, mt_worker, &job) == 0)
            started++;
        else
            break;
    if (started == 0)
        mt_worker(&job);
    for (int i = 0; i < started; i++)
        pthread_join(th[i], NULL);
    free(th);
    return (long)atomic_load(&job.total);
}
