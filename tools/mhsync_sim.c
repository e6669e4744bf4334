// CPU model of the sub-sequence synchronisation of one image (design study for the small-job path of k_sync):
//   (A) the round-based scheme of k_sync: speculative tail walk, then "walk again from the left neighbour's exit state" until nothing
//       changes -- rounds per workgroup of 255 sub-sequences (the serial chain a single image waits for);
//   (B) the candidate scheme: one speculative walk per block-in-MCU index (phase) from one sub-sequence further left, one walk of the
//       own sub-sequence from every candidate entry state, then the chain of selections -- how many sub-sequences find their true entry
//       state among the candidates, and how many rounds of (A) are left for the others.
//   gcc -O2 -Ioracle -o /tmp/mhsync_sim tools/mhsync_sim.c oracle/jpeg_synth.c -lm && /tmp/mhsync_sim 3840 2160 77 512
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "jpeg_synth.h"
typedef struct { uint16_t lut[65536]; } Tab;
static void build(Tab* t, const uint8_t* counts, const uint8_t* vals)
{
    memset(t, 0, sizeof *t); unsigned code = 0, k = 0;
    for (int l = 1; l <= 16; l++) { for (int i = 0; i < counts[l - 1]; i++, k++) { unsigned lo = code << (16 - l), n = 1u << (16 - l); for (unsigned j = 0; j < n; j++) t->lut[lo + j] = (l << 8) | vals[k]; code++; } code <<= 1; }
}
static uint8_t* U; static size_t UL, TB;
static inline unsigned peek16(size_t p) { size_t b = p >> 3; unsigned v = ((unsigned)U[b] << 24) | (U[b + 1] << 16) | (U[b + 2] << 8) | U[b + 3]; return (v << (p & 7)) >> 16; }
typedef struct { size_t p; int c, k; } St;
static Tab T[4]; static int nb; static int slot_of[10];
static inline int st_eq(St a, St b) { return a.p == b.p && a.c == b.c && a.k == b.k; }
static void step(St* s)
{
    Tab* t = &T[slot_of[s->c] * 2 + (s->k ? 1 : 0)]; unsigned e = t->lut[peek16(s->p)]; unsigned len = e >> 8, sym = e & 255; if (!len) { s->p += 1; return; }
    s->p += len + (sym & 15); int done = 0; if (s->k == 0) s->k = 1; else if (sym == 0) done = 1; else { s->k += (sym >> 4) + 1; if (s->k >= 64) done = 1; }
    if (done) { s->k = 0; s->c = (s->c + 1) % nb; }
}
static St walk(St s, size_t end) { if (end > TB) end = TB; while (s.p < end) step(&s); return s; }

int main(int argc, char** argv)
{
    int W = argc > 1 ? atoi(argv[1]) : 3840, H = argc > 2 ? atoi(argv[2]) : 2160, seed = argc > 3 ? atoi(argv[3]) : 77, S = argc > 4 ? atoi(argv[4]) : 512;
    int heur = argc > 9 ? atoi(argv[9]) : 1; int hs = argc > 5 ? atoi(argv[5]) : 2, vs = argc > 6 ? atoi(argv[6]) : 2, noise = argc > 7 ? atoi(argv[7]) : 12, tail = argc > 8 ? atoi(argv[8]) : 1024;
    JsynthParams p = { W, H, hs, vs, 85, 0, 0, 0, 0, noise, (uint32_t)seed }; size_t cap = (size_t)W * H * 3 + 65536; uint8_t* f = malloc(cap); size_t n = jsynth_encode(&p, f, cap);
    size_t pos = 2, ss = 0; uint8_t cnt[4][16], val[4][256];
    while (pos < n) { unsigned m = f[pos + 1]; unsigned len = (f[pos + 2] << 8) | f[pos + 3];
        if (m == 0xC4) { size_t q = pos + 4; while (q < pos + 2 + len) { int tc = f[q] >> 4, th = f[q] & 15; int id = th * 2 + tc; memcpy(cnt[id], f + q + 1, 16); int tot = 0; for (int i = 0; i < 16; i++) tot += cnt[id][i]; memcpy(val[id], f + q + 17, tot); q += 17 + tot; } }
        if (m == 0xDA) { ss = pos + 2 + len; break; } pos += 2 + len; }
    for (int i = 0; i < 4; i++) build(&T[i], cnt[i], val[i]);
    nb = hs * vs + 2; for (int i = 0; i < hs * vs; i++) slot_of[i] = 0; slot_of[hs * vs] = 1; slot_of[hs * vs + 1] = 1;
    U = malloc(n + 16); UL = 0; for (size_t i = ss; i + 1 < n; i++) { if (f[i] == 0xFF && f[i + 1] == 0xD9) break; U[UL++] = f[i]; if (f[i] == 0xFF && f[i + 1] == 0) i++; } memset(U + UL, 0, 16);
    TB = UL * 8; size_t nsub = (TB + S - 1) / S;
    St* truth = malloc(sizeof(St) * (nsub + 1)); { St s = { 0, 0, 0 }; for (size_t i = 0; i < nsub; i++) { s = walk(s, (i + 1) * (size_t)S); truth[i] = s; } }
    printf("%dx%d seed %d: scan %zu bytes, %zu sub-sequences of %d bits, %d blocks per MCU\n", W, H, seed, UL, nsub, S, nb);

    // ---- (A) rounds of the present scheme, per workgroup of 255 owned sub-sequences (+ a speculative halo walk of the one before)
    {
        St* out = malloc(sizeof(St) * nsub); St* in = malloc(sizeof(St) * nsub); long max_it = 0, sum_it = 0, nwg = 0; long walks = 0; size_t wrong_after1 = 0;
        St* pin = malloc(sizeof(St) * nsub); St* pout = malloc(sizeof(St) * nsub); for (size_t i = 0; i < nsub; i++) { pin[i] = (St){ (size_t)-3, 0, 0 }; pout[i] = pin[i]; }
        const long rmax = getenv("SIM_RMAX") ? atol(getenv("SIM_RMAX")) : 1000000;
        for (size_t g0 = 0; g0 < nsub; g0 += 255, nwg++) {
            size_t g1 = g0 + 255 < nsub ? g0 + 255 : nsub;
            St halo = { 0, 0, 0 }; if (g0) { size_t i = g0 - 1; long sp = (long)i * S + (S - tail); if (sp < 0) sp = 0; St s = { (size_t)sp, 0, 0 }; halo = walk(s, (i + 1) * (size_t)S); }
            for (size_t i = g0; i < g1; i++) { if (i == 0) { St s = { 0, 0, 0 }; in[i] = s; out[i] = walk(s, S); } else { long sp = (long)i * S + (S - tail); St s = { (size_t)sp, 0, 0 }; in[i] = s; out[i] = walk(s, (i + 1) * (size_t)S); } walks++; }
            long it = 1;
            for (;; it++) {
                int changed = 0; St* nin = malloc(sizeof(St) * (g1 - g0));
                for (size_t i = g0; i < g1; i++) nin[i - g0] = i == 0 ? in[0] : (i == g0 ? halo : out[i - 1]);
                if (it > rmax) { free(nin); break; }
                for (size_t i = g0; i < g1; i++) if (!st_eq(nin[i - g0], in[i])) { pin[i] = in[i]; pout[i] = out[i]; in[i] = nin[i - g0]; St o = in[i].p >= (i + 1) * (size_t)S ? in[i] : walk(in[i], (i + 1) * (size_t)S); walks++; if (!st_eq(o, out[i])) { if (getenv("SIM_TRACE2") && g0 == (size_t)atoi(getenv("SIM_TRACE2"))) printf("    round %ld sub %zu: exit (%zu,%d,%d) -> (%zu,%d,%d) truth (%zu,%d,%d)\n", it, i - g0, out[i].p, out[i].c, out[i].k, o.p, o.c, o.k, truth[i].p, truth[i].c, truth[i].k); out[i] = o; changed = 1; } }
                free(nin); if (!changed) break;
            }
            if (getenv("SIM_TRACE") && it >= atoi(getenv("SIM_TRACE"))) { printf("  wg at %zu: %ld rounds; final wrong:", g0, it); for (size_t i = g0; i < g1; i++) if (!st_eq(out[i], truth[i])) printf(" %zu", i - g0); printf("\n"); }
            if (it > max_it) max_it = it; sum_it += it;
        }
        for (size_t i = 0; i < nsub; i++) if (!st_eq(out[i], truth[i])) wrong_after1++;
        printf("(A) rounds per workgroup in the first launch: max %ld, mean %.1f over %ld workgroups; %ld walks (%.2f per sub-sequence); %zu exits still wrong after it\n", max_it, (double)sum_it / nwg, nwg, walks, (double)walks / nsub, wrong_after1);
        if (getenv("SIM_RMAX")) {
            // ---- (D) the rounds stopped after SIM_RMAX of them; every sub-sequence keeps its last two walks as a memo; the chain of look-ups + queued walks as in (C)
            enum { E = 16 }; St* me = malloc(sizeof(St) * nsub * E); St* mx = malloc(sizeof(St) * nsub * E); int* mn = calloc(nsub, sizeof(int));
            size_t open = 0; for (size_t i = 1; i < nsub; i++) if (!st_eq(in[i], out[i - 1])) open++;
            for (size_t i = 0; i < nsub; i++) { me[i * E] = in[i]; mx[i * E] = out[i]; mn[i] = 1; if (pin[i].p != (size_t)-3 && !st_eq(pin[i], in[i])) { me[i * E + 1] = pin[i]; mx[i * E + 1] = pout[i]; mn[i] = 2; } }
            long iters = 0, total_walks = 0, max_req = 0;
            for (;; iters++) {
                size_t* req_i = malloc(sizeof(size_t) * nsub); St* req_s = malloc(sizeof(St) * nsub); long nreq = 0;
                St cur = { 0, 0, 0 }; int have = 1;
                for (size_t i = 0; i < nsub; i++) {
                    int f = -1; if (have) for (int q = 0; q < mn[i]; q++) if (st_eq(me[i * E + q], cur)) { f = q; break; }
                    if (f >= 0) { cur = mx[i * E + f]; continue; }
                    if (have) { req_i[nreq] = i; req_s[nreq] = cur; nreq++; }
                    cur = mx[i * E]; have = 1;                    // guess: the exit of the latest walk
                }
                if (!nreq) { free(req_i); free(req_s); break; }
                for (long r = 0; r < nreq; r++) { size_t i = req_i[r]; if (mn[i] < E) { me[i * E + mn[i]] = req_s[r]; mx[i * E + mn[i]] = req_s[r].p >= (i + 1) * (size_t)S ? req_s[r] : walk(req_s[r], (i + 1) * (size_t)S); mn[i]++; } }
                total_walks += nreq; if (nreq > max_req) max_req = nreq; free(req_i); free(req_s);
                if (iters > 200) break;
            }
            St cur = { 0, 0, 0 }; size_t bad = 0; for (size_t i = 0; i < nsub; i++) { int f = -1; for (int q = 0; q < mn[i]; q++) if (st_eq(me[i * E + q], cur)) { f = q; break; } if (f < 0) { bad++; break; } cur = mx[i * E + f]; if (!st_eq(cur, truth[i])) bad++; }
            printf("(D) %ld rounds, then %zu open links; memo chain: %ld extra walk rounds, %ld walks in them (at most %ld in one), chain %s\n", rmax, open, iters, total_walks, max_req, bad ? "WRONG" : "= truth");
        }
    }
    // ---- (B) candidates
    {
        St* X = malloc(sizeof(St) * nsub * nb); St* Y = malloc(sizeof(St) * nsub * nb);
        for (size_t i = 0; i < nsub; i++) for (int h = 0; h < nb; h++) {
            if (i == 0) { St s = { 0, 0, 0 }; X[h] = walk(s, S); continue; }
            int hh = h; if (getenv("SIM_HYPS")) { const char* e = getenv("SIM_HYPS"); int ok = 0; for (const char* q = e; *q; q++) if (*q - '0' == h) ok = 1; if (!ok) hh = e[0] - '0'; }
            long sp = (long)i * S + (S - tail); if (sp < 0) sp = 0; St s = { (size_t)sp, hh, 0 }; X[i * nb + h] = walk(s, (i + 1) * (size_t)S);
        }
        for (size_t i = 0; i < nsub; i++) for (int h = 0; h < nb; h++) {
            if (i == 0) { Y[h] = X[h]; continue; }
            St e = X[(i - 1) * nb + h]; Y[i * nb + h] = e.p >= (i + 1) * (size_t)S ? e : walk(e, (i + 1) * (size_t)S);
        }
        // chain: sel[i] = candidate of sub-sequence i that is its true entry state (-1: none)
        int* sel = malloc(sizeof(int) * nsub); sel[0] = 0; size_t none = 0, bridged = 0, wrong = 0, distinct_sum = 0;
        for (size_t i = 0; i + 1 < nsub; i++) {
            int nx = -1;
            if (sel[i] >= 0) { St o = Y[i * nb + sel[i]]; for (int h = 0; h < nb; h++) if (st_eq(X[i * nb + h], o)) { nx = h; break; } }
            if (heur && nx < 0) {                                // no proven successor: (1) every speculative walk of i ended in the same state; (2) the exit most walks from the candidates agree on
                int all = 1; for (int h = 1; h < nb; h++) if (!st_eq(X[i * nb + h], X[i * nb])) all = 0;
                if (all) nx = 0;
                else { int best = 0; for (int h = 0; h < nb; h++) { int cnt = 0; for (int g = 0; g < nb; g++) if (st_eq(Y[i * nb + g], X[i * nb + h])) cnt++; if (cnt > best) { best = cnt; nx = h; } } }
                if (nx >= 0) bridged++;
            }
            sel[i + 1] = nx;
        }
        { size_t pk = 0, nopk = 0, first = 0; for (size_t i = 1; i < nsub; i++) { int m = 0, mpk = 0; for (int h = 0; h < nb; h++) { St x = X[(i - 1) * nb + h]; if (st_eq(x, truth[i - 1])) m = 1; if (x.p == truth[i - 1].p && x.k == truth[i - 1].k) mpk = 1; } if (!m) { if (mpk) pk++; else nopk++; } }
          printf("    truth missing among the candidates of %zu sub-sequences with position and index right (phase only), of %zu with nothing right\n", pk, nopk); (void)first; }
        for (size_t i = 0; i < nsub; i++) { if (sel[i] < 0) none++; else if (!st_eq(Y[i * nb + sel[i]], truth[i])) wrong++;
            int d = 0; for (int h = 0; h < nb; h++) { int dup = 0; for (int g = 0; g < h; g++) if (i && st_eq(X[(i - 1) * nb + g], X[(i - 1) * nb + h])) dup = 1; if (!dup) d++; } distinct_sum += d; }
        printf("(B) %zu of %zu sub-sequences without their true entry among the %d candidates (%zu bridged over a consensus), %zu selected exits differ from the truth; %.2f distinct candidates per sub-sequence\n",
               none, nsub, nb, bridged, wrong, (double)distinct_sum / nsub);
        // ---- (C) the same candidates as a memo (entry state -> exit state) per sub-sequence; iterate { follow the chain from the true start by
        //      look-ups; where it meets a sub-sequence that was never walked from that state: note it, guess a way on (consensus / most common
        //      exit), keep following; walk everything noted in ONE round, add to the memos } until the chain runs through
        {
            enum { E = 16 }; St* me = malloc(sizeof(St) * nsub * E); St* mx = malloc(sizeof(St) * nsub * E); int* mn = calloc(nsub, sizeof(int));
            for (size_t i = 0; i < nsub; i++) for (int h = 0; h < nb; h++) { St e = i ? X[(i - 1) * nb + h] : (St){ 0, 0, 0 }; int dup = 0; for (int q = 0; q < mn[i]; q++) if (st_eq(me[i * E + q], e)) dup = 1; if (!dup) { me[i * E + mn[i]] = e; mx[i * E + mn[i]] = Y[i * nb + h]; mn[i]++; } }
            long iters = 0, total_walks = 0, max_req = 0; long* filled = calloc(nsub, sizeof(long));
            for (;; iters++) {
                size_t* req_i = malloc(sizeof(size_t) * nsub); St* req_s = malloc(sizeof(St) * nsub); long nreq = 0;
                St cur = { 0, 0, 0 }; int have = 1;
                for (size_t i = 0; i < nsub; i++) {
                    int f = -1; if (have) for (int q = 0; q < mn[i]; q++) if (st_eq(me[i * E + q], cur)) { f = q; break; }
                    if (f >= 0) { cur = mx[i * E + f]; continue; }
                    if (have) { req_i[nreq] = i; req_s[nreq] = cur; nreq++; }
                    // guess the exit of i: every speculative walk agrees, else the exit most memo entries lead to
                    have = 0; int all = 1; for (int h = 1; h < nb; h++) if (!st_eq(X[i * nb + h], X[i * nb])) all = 0;
                    if (all) { cur = X[i * nb]; have = 1; }
                    else if (getenv("SIM_GUESS_X")) { int best = 0; for (int q = 0; q < nb; q++) { int c2 = 0; for (int r = 0; r < nb; r++) if (st_eq(X[i * nb + r], X[i * nb + q])) c2++; if (c2 > best && c2 >= atoi(getenv("SIM_GUESS_X"))) { best = c2; cur = X[i * nb + q]; have = 1; } } }
                    else { int best = 0; for (int q = 0; q < mn[i]; q++) { int c2 = 0; for (int r = 0; r < mn[i]; r++) if (st_eq(mx[i * E + r], mx[i * E + q])) c2++; if (c2 > best && c2 >= 2) { best = c2; cur = mx[i * E + q]; have = 1; } } }
                }
                if (!nreq) { free(req_i); free(req_s); break; }
                if (iters >= 1 && getenv("SIM_TRACE3")) { long dbl = 0; for (long r = 0; r < nreq; r++) { size_t i = req_i[r]; if (i && mn[i - 1] > 0 && st_eq(mx[(i - 1) * E + mn[i - 1] - 1], req_s[r]) && filled[i - 1] == iters) dbl++; } printf("    round %ld: %ld requests, %ld of them right behind a sub-sequence filled in the round before\n", iters + 1, nreq, dbl); }
                for (long r = 0; r < nreq; r++) { size_t i = req_i[r]; filled[i] = iters + 1; if (mn[i] < E) { me[i * E + mn[i]] = req_s[r]; mx[i * E + mn[i]] = req_s[r].p >= (i + 1) * (size_t)S ? req_s[r] : walk(req_s[r], (i + 1) * (size_t)S); mn[i]++; } }
                total_walks += nreq; if (nreq > max_req) max_req = nreq; free(req_i); free(req_s);
                if (iters > 200) break;
            }
            St cur = { 0, 0, 0 }; size_t bad = 0; for (size_t i = 0; i < nsub; i++) { int f = -1; for (int q = 0; q < mn[i]; q++) if (st_eq(me[i * E + q], cur)) { f = q; break; } if (f < 0) { bad++; break; } cur = mx[i * E + f]; if (!st_eq(cur, truth[i])) bad++; }
            int maxm = 0; for (size_t i = 0; i < nsub; i++) if (mn[i] > maxm) maxm = mn[i];
            printf("(C) memo iteration: %ld extra walk rounds, %ld walks in them (at most %ld in one), largest memo %d, chain %s\n", iters, total_walks, max_req, maxm, bad ? "WRONG" : "= truth");
        }
        // rounds of (A) left: start from the selected states, iterate per workgroup
        St* out = malloc(sizeof(St) * nsub); St* in = malloc(sizeof(St) * nsub);
        for (size_t i = 0; i < nsub; i++) { if (sel[i] >= 0) { in[i] = i ? X[(i - 1) * nb + sel[i]] : (St){ 0, 0, 0 }; out[i] = Y[i * nb + sel[i]]; } else { in[i] = (St){ (size_t)-2, 0, 0 }; out[i] = Y[i * nb]; } }
        long max_it = 0, walks = 0; size_t wrong1 = 0;
        for (size_t g0 = 0; g0 < nsub; g0 += 255) {
            size_t g1 = g0 + 255 < nsub ? g0 + 255 : nsub; long it = 0;
            St halo = g0 ? out[g0 - 1] : (St){ 0, 0, 0 };
            for (;; it++) {
                int changed = 0; St* nin = malloc(sizeof(St) * (g1 - g0));
                for (size_t i = g0; i < g1; i++) nin[i - g0] = i == 0 ? (St){ 0, 0, 0 } : (i == g0 ? halo : out[i - 1]);
                for (size_t i = g0; i < g1; i++) if (!st_eq(nin[i - g0], in[i])) { in[i] = nin[i - g0]; St o = in[i].p >= (i + 1) * (size_t)S ? in[i] : walk(in[i], (i + 1) * (size_t)S); walks++; if (!st_eq(o, out[i])) { out[i] = o; changed = 1; } }
                free(nin); if (!changed) break;
            }
            if (it > max_it) max_it = it;
        }
        for (size_t i = 0; i < nsub; i++) if (!st_eq(out[i], truth[i])) wrong1++;
        printf("    verification launch: max %ld rounds in a workgroup, %ld walks, %zu exits wrong after it (left to the boundary launch)\n", max_it, walks, wrong1);
    }
    return 0;
}
