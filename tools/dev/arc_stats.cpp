/* arc_stats.cpp — DEV ONLY: statistics of the Lee-Richards work on synthetic inputs, used to size
 * kernel design choices on the CPU (lane balance of the arc pass, culling rates, candidate counts).
 *   g++ -O2 -o /tmp/arc_stats tools/dev/arc_stats.cpp tools/synth.c -lm && /tmp/arc_stats coil 10000 4
 * Not part of the product, not part of the oracle. */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

extern "C" int synth_coil(int n, uint64_t seed, double *xyz, double *radii);
extern "C" int synth_globule(int n, uint64_t seed, double spacing, double *xyz, double *radii);

struct Pair { int j; double beta, xd, yd, zd, d3sq, dij, costheta; bool culled; };

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "coil";
    const int n = argc > 2 ? atoi(argv[2]) : 10000;
    const int nstruct = argc > 3 ? atoi(argv[3]) : 2;
    const int ns = 20;
    const double probe = 1.4;
    const int KMAX = 5; /* container counts tested: 0,1,2,4,8 */
    const int Ks[KMAX] = {0, 1, 2, 4, 8};
    double sum_nn = 0, sum_arcs[KMAX] = {0}, sum_tilemax[KMAX] = {0}, sum_nnmax[KMAX] = {0}, culled[KMAX] = {0};
    double sum_gaps = 0, sum_slices = 0, sum_buried = 0, sum_tiles = 0, sum_pairs = 0;
    double sum_cand = 0, sum_cand_half = 0, sum_groups = 0;
    double sum_split6 = 0, sum_plain6 = 0, sum_lane6 = 0, sum_ideal6 = 0, tiles6 = 0, sum_tilemax_gaps = 0;
    double sum_pairs_tile_rounds = 0;
    static double qsim_iters[2][3][4][2], qsim_events[2][3][4][2], qsim_tiles[2][3][4][2];
    double sk_arcs = 0, sk_exact = 0, sk_ub = 0, sk_exit = 0, sk_exit_ub = 0; /* arcs the union could skip */
    double hist[64] = {0}, ghist[8] = {0}, sum_tilemax_open = 0, sum_tilemax_top2 = 0, arcs_open = 0;
    for (int st = 0; st < nstruct; ++st) {
        std::vector<double> xyz(3 * n), rad(n);
        if (!strcmp(kind, "coil")) synth_coil(n, 1000 + st, xyz.data(), rad.data());
        else synth_globule(n, 500 + st, 2.6, xyz.data(), rad.data());
        std::vector<double> R(n);
        double rmax = 0, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; ++i) {
            R[i] = rad[i] + probe; rmax = std::max(rmax, R[i]);
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], xyz[3 * i + k]); hi[k] = std::max(hi[k], xyz[3 * i + k]); }
        }
        const double d = 2 * rmax;
        int nc[3];
        for (int k = 0; k < 3; ++k) { lo[k] -= d / 2; nc[k] = (int)ceil((hi[k] + d / 2 - lo[k]) / d); }
        std::vector<long long> cell(n);
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) {
            int c[3];
            for (int k = 0; k < 3; ++k) c[k] = (int)((xyz[3 * i + k] - lo[k]) / d);
            cell[i] = c[0] + (long long)nc[0] * (c[1] + (long long)nc[1] * c[2]);
            order[i] = i;
        }
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cell[a] < cell[b]; });
        std::map<long long, int> occ;
        for (int i = 0; i < n; ++i) occ[cell[i]]++;
        /* half-size cells for the candidate estimate */
        std::map<long long, int> occ_half;
        auto hkey = [&](int i, int dx, int dy, int dz) {
            long long c[3];
            const int dd[3] = {dx, dy, dz};
            for (int k = 0; k < 3; ++k) c[k] = (long long)((xyz[3 * i + k] - lo[k]) / (d / 2)) + dd[k];
            return c[0] + 4096 * (c[1] + 4096 * c[2]);
        };
        for (int i = 0; i < n; ++i) occ_half[hkey(i, 0, 0, 0)]++;

        /* neighbor lists, brute force over the 27 cells */
        std::map<long long, std::vector<int>> members;
        for (int i = 0; i < n; ++i) members[cell[i]].push_back(i);
        std::vector<std::vector<Pair>> nb(n);
        for (int i = 0; i < n; ++i) {
            int c[3];
            for (int k = 0; k < 3; ++k) c[k] = (int)((xyz[3 * i + k] - lo[k]) / d);
            int cand = 0;
            for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
                if (x < 0 || y < 0 || z < 0 || x >= nc[0] || y >= nc[1] || z >= nc[2]) continue;
                auto it = members.find(x + (long long)nc[0] * (y + (long long)nc[1] * z));
                if (it == members.end()) continue;
                for (int j : it->second) {
                    ++cand;
                    if (j == i) continue;
                    const double xd = xyz[3 * j] - xyz[3 * i], yd = xyz[3 * j + 1] - xyz[3 * i + 1], zd = xyz[3 * j + 2] - xyz[3 * i + 2];
                    const double d3 = xd * xd + yd * yd + zd * zd, cut = R[i] + R[j];
                    if (d3 < cut * cut) {
                        Pair p; p.j = j; p.xd = xd; p.yd = yd; p.zd = zd; p.d3sq = d3; p.dij = sqrt(xd * xd + yd * yd);
                        p.beta = atan2(yd, xd) + M_PI;
                        p.costheta = (R[i] * R[i] + d3 - R[j] * R[j]) / (2 * R[i] * sqrt(d3));
                        p.culled = false;
                        nb[i].push_back(p);
                    }
                }
            }
            sum_cand += cand;
            int ch = 0;
            for (int dz = -2; dz <= 2; ++dz) for (int dy = -2; dy <= 2; ++dy) for (int dx = -2; dx <= 2; ++dx) {
                auto it = occ_half.find(hkey(i, dx, dy, dz));
                if (it != occ_half.end()) ch += it->second;
            }
            sum_cand_half += ch;
            std::sort(nb[i].begin(), nb[i].end(), [](const Pair &a, const Pair &b) { return a.beta < b.beta; });
            sum_nn += nb[i].size();
        }
        for (int kc = 0; kc < KMAX; ++kc) {
            const int K = Ks[kc];
            std::vector<std::vector<int>> arcs(n, std::vector<int>(ns, 0));
            std::vector<std::vector<int>> gaps(n, std::vector<int>(ns, 0));
            std::vector<int> nnk(n);
            for (int i = 0; i < n; ++i) {
                auto &L = nb[i];
                for (auto &p : L) p.culled = false;
                if (K > 0) {
                    std::vector<int> idx(L.size());
                    for (size_t k = 0; k < L.size(); ++k) idx[k] = (int)k;
                    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return L[a].costheta < L[b].costheta; });
                    for (int c = 0; c < K && c < (int)L.size(); ++c) {
                        const Pair &C = L[idx[c]];
                        const double thC = acos(std::max(-1.0, std::min(1.0, C.costheta)));
                        for (size_t k = 0; k < L.size(); ++k) {
                            if ((int)k == idx[c] || L[k].culled) continue;
                            const Pair &P = L[k];
                            const double thP = acos(std::max(-1.0, std::min(1.0, P.costheta)));
                            const double cg = (P.xd * C.xd + P.yd * C.yd + P.zd * C.zd) / sqrt(P.d3sq * C.d3sq);
                            const double g = acos(std::max(-1.0, std::min(1.0, cg)));
                            if (g + thP <= thC) L[k].culled = true;
                        }
                    }
                }
                int kept = 0;
                for (auto &p : L) if (!p.culled) ++kept;
                nnk[i] = kept;
                culled[kc] += L.size() - kept;
                const double Ri = R[i], delta = 2 * Ri / ns;
                for (int s = 0; s < ns; ++s) {
                    const double t = -Ri - 0.5 * delta + (s + 1) * delta;
                    const double A = Ri * Ri - t * t;
                    if (!(A > 0)) continue;
                    const double Rip = sqrt(A);
                    bool buried = false;
                    std::vector<std::pair<double, double>> as;
                    std::vector<double> cs, bs;
                    for (auto &p : L) {
                        if (p.culled) continue;
                        const double Kp = Ri * Ri - R[p.j] * R[p.j] + p.d3sq;
                        const double c = (Kp - 2 * p.zd * t) / (2 * Rip * p.dij);
                        if (c >= 1) continue;
                        if (c <= -1) { buried = true; break; }
                        const double al = acos(c);
                        as.push_back({p.beta - al, p.beta + al});
                        cs.push_back(c); bs.push_back(p.beta);
                    }
                    if (buried) { if (kc == 0) sum_buried += 1; continue; }
                    arcs[i][s] = (int)as.size();
                    if (kc == 0) {
                        /* what a containment test before the acos could skip: the beta-ordered stack union */
                        std::vector<std::pair<double, double>> stk;
                        bool done = false, done_ub = false;
                        for (size_t k = 0; k < as.size(); ++k) {
                            sk_arcs += 1;
                            if (done) sk_exit += 1;
                            const double c = cs[k];
                            const double ub = c < 0 ? 1.5707963267948966 * (1 - c)
                                : std::min(acos(0.3) - (c - 0.3) / sqrt(1 - 0.09), acos(0.85) - (c - 0.85) / sqrt(1 - 0.85 * 0.85));
                            if (!stk.empty()) {
                                const double m = std::min(bs[k] - stk.back().first, stk.back().second - bs[k]);
                                if (as[k].first >= stk.back().first && as[k].second <= stk.back().second) sk_exact += 1;
                                if (ub <= m) { sk_ub += 1; if (!done) { if (done_ub) sk_exit_ub += 0; } }
                                else if (done) sk_exit_ub += 1; /* skipped by the exit but not by the bound */
                            }
                            if (stk.empty() || as[k].first > stk.back().second) stk.push_back(as[k]);
                            else {
                                stk.back().first = std::min(stk.back().first, as[k].first);
                                stk.back().second = std::max(stk.back().second, as[k].second);
                                while (stk.size() > 1 && stk[stk.size() - 2].second >= stk.back().first) {
                                    stk[stk.size() - 2].first = std::min(stk[stk.size() - 2].first, stk.back().first);
                                    stk[stk.size() - 2].second = stk.back().second;
                                    stk.pop_back();
                                }
                            }
                            if (stk.size() == 1 && stk[0].second - stk[0].first >= 2 * M_PI) done = true;
                        }
                        (void)done_ub;
                        /* gaps of the union (on the circle) */
                        std::vector<std::pair<double, double>> iv;
                        for (auto &a : as) {
                            double lo = a.first, hi = a.second;
                            if (lo < 0) { iv.push_back({lo + 2 * M_PI, 2 * M_PI}); lo = 0; }
                            if (hi > 2 * M_PI) { iv.push_back({0, hi - 2 * M_PI}); hi = 2 * M_PI; }
                            iv.push_back({lo, hi});
                        }
                        std::sort(iv.begin(), iv.end());
                        int g = 0; double sup = 0;
                        for (auto &v : iv) { if (v.first > sup) ++g; sup = std::max(sup, v.second); }
                        if (sup < 2 * M_PI) ++g;
                        gaps[i][s] = g;
                        ghist[std::min(g, 7)] += 1;
                        if (g > 0) arcs_open += as.size();
                        sum_gaps += g;
                        hist[std::min<int>(63, as.size())] += 1;
                    }
                }
                if (kc == 0) sum_slices += ns;
            }
            /* tiles of 3 consecutive cell-sorted atoms */
            for (int t0 = 0; t0 + 3 <= n; t0 += 3) {
                int mx = 0, nnmx = 0, gmx = 0, tot = 0;
                for (int a = 0; a < 3; ++a) {
                    const int i = order[t0 + a];
                    nnmx = std::max(nnmx, (nnk[i] + 1) & ~1);
                    tot += (nnk[i] + 1) & ~1;
                    for (int s = 0; s < ns; ++s) { mx = std::max(mx, arcs[i][s]); sum_arcs[kc] += arcs[i][s]; gmx = std::max(gmx, gaps[i][s]); }
                }
                sum_tilemax[kc] += mx; sum_nnmax[kc] += nnmx;
                if (kc == 0) { int mo = 0; std::vector<int> all; for (int a = 0; a < 3; ++a) for (int s = 0; s < ns; ++s) { all.push_back(arcs[order[t0 + a]][s]); if (gaps[order[t0 + a]][s] > 0) mo = std::max(mo, arcs[order[t0 + a]][s]); }
                    std::sort(all.begin(), all.end(), std::greater<int>()); sum_tilemax_open += mo; sum_tilemax_top2 += all[4];
                    sum_tiles += 1; sum_tilemax_gaps += gmx; sum_pairs += tot; sum_pairs_tile_rounds += (tot + 63) / 64;
                    int groups = 1;
                    for (int a = 1; a < 3; ++a) if (cell[order[t0 + a]] != cell[order[t0 + a - 1]]) ++groups;
                    sum_groups += groups; }
            }
            if (kc == 0 || kc == 3) {
                const int TAs[3] = {4, 5, 6};
                const int Ts[4] = {8, 16, 24, 32};
                for (int ti = 0; ti < 3; ++ti) for (int th = 0; th < 4; ++th) for (int sorted = 0; sorted < 2; ++sorted) {
                    const int TA = TAs[ti], T = Ts[th];
                    for (int t0 = 0; t0 + TA <= n; t0 += TA) {
                        std::vector<int> q;
                        for (int a = 0; a < TA; ++a) for (int s2 = 0; s2 < ns; ++s2) q.push_back(arcs[order[t0 + a]][s2]);
                        if (sorted) std::sort(q.begin(), q.end(), std::greater<int>());
                        const int ni = (int)q.size();
                        int rem[64]; bool has[64];
                        int next = 0, iters = 0, events = 0;
                        for (int l = 0; l < 64; ++l) { has[l] = next < ni; rem[l] = has[l] ? q[next++] : 0; }
                        for (;;) {
                            int idle = 0, busy = 0;
                            for (int l = 0; l < 64; ++l) { if (rem[l] == 0) ++idle; else ++busy; }
                            if (next < ni && (idle >= T || busy == 0)) {
                                ++events;
                                for (int l = 0; l < 64 && next < ni; ++l) if (rem[l] == 0) { rem[l] = q[next++]; }
                                continue;
                            }
                            if (busy == 0) break;
                            ++iters;
                            for (int l = 0; l < 64; ++l) if (rem[l] > 0) --rem[l];
                        }
                        qsim_iters[kc == 3][ti][th][sorted] += iters; qsim_events[kc == 3][ti][th][sorted] += events; qsim_tiles[kc == 3][ti][th][sorted] += 1;
                    }
                }
            }
            if (kc == 0)
                for (int t0 = 0; t0 + 6 <= n; t0 += 6) {
                    std::vector<int> c;
                    int m1 = 0, m2 = 0;
                    for (int a = 0; a < 6; ++a) for (int s = 0; s < ns; ++s) {
                        const int v = arcs[order[t0 + a]][s];
                        c.push_back(v);
                        if (a < 3) m1 = std::max(m1, v); else m2 = std::max(m2, v);
                    }
                    { int mh = 0, ml = 0; double tot = 0; for (int l = 0; l < 60; ++l) { mh = std::max(mh, std::max(c[l], c[l + 60])); ml = std::max(ml, std::min(c[l], c[l + 60])); tot += c[l] + c[l + 60]; }
                      sum_lane6 += mh + ml; sum_ideal6 += tot / 64.0; }
                    std::sort(c.begin(), c.end(), std::greater<int>());
                    sum_split6 += c[0] + c[64];
                    sum_plain6 += m1 + m2;
                    tiles6 += 1;
                }
        }
    }
    const double N = (double)n * nstruct;
    printf("%s n=%d x %d: nn/atom %.2f  candidates/atom %.1f (half-size cells: %.1f)  cell groups per 3-atom tile %.2f\n", kind, n, nstruct,
           sum_nn / N, sum_cand / N, sum_cand_half / N, sum_groups / sum_tiles);
    printf("slices: buried %.3f, gaps/slice %.2f, max gaps per tile %.2f; pairs(padded)/tile %.1f, pair rounds/tile %.2f\n", sum_buried / sum_slices, sum_gaps / sum_slices,
           sum_tilemax_gaps / sum_tiles, sum_pairs / sum_tiles, sum_pairs_tile_rounds / sum_tiles);
    for (int kc = 0; kc < KMAX; ++kc)
        printf("containers %d: culled %.1f%% of neighbors; arcs/slice %.2f; tile max arcs (iterations) %.2f; tile max nn (padded) %.2f\n", Ks[kc],
               100 * culled[kc] / sum_nn, sum_arcs[kc] / (sum_tiles * 60), sum_tilemax[kc] / sum_tiles, sum_nnmax[kc] / sum_tiles);
    printf("6-atom tiles: plain two rounds %.2f iterations, heavy/light split %.2f, per-lane heavy/light %.2f, perfect balance %.2f\n", sum_plain6 / tiles6, sum_split6 / tiles6, sum_lane6 / tiles6, sum_ideal6 / tiles6);
    printf("containment before acos: arcs %.0f, exactly inside the top component %.3f, linear bound says so %.3f, after full coverage (exit) %.3f, exit-only on top of bound %.3f\n", sk_arcs, sk_exact / sk_arcs, sk_ub / sk_arcs, sk_exit / sk_arcs, sk_exit_ub / sk_arcs);
    printf("gap count histogram:"); for (int k = 0; k < 8; ++k) printf(" %d:%.3f", k, ghist[k] / sum_slices);
    printf("\ntile max arcs over slices with >=1 gap: %.2f; 5th largest lane: %.2f; arcs in open slices %.2f of all\n", sum_tilemax_open / sum_tiles, sum_tilemax_top2 / sum_tiles, arcs_open / sum_arcs[0]);
    for (int c = 0; c < 2; ++c) for (int ti = 0; ti < 3; ++ti) for (int th = 0; th < 4; ++th) for (int so = 0; so < 2; ++so)
        printf("queue sim containers=%d TA=%d T=%d sorted=%d: arc iterations per 3 atoms %.2f, refill events per 3 atoms %.2f\n", c ? 4 : 0, (int[]){4,5,6}[ti], (int[]){8,16,24,32}[th], so,
               qsim_iters[c][ti][th][so] / qsim_tiles[c][ti][th][so] * 3 / (int[]){4,5,6}[ti], qsim_events[c][ti][th][so] / qsim_tiles[c][ti][th][so] * 3 / (int[]){4,5,6}[ti]);
    printf("arc count histogram:");
    for (int k = 0; k < 40; ++k) printf(" %d:%.3f", k, hist[k] / sum_slices);
    printf("\n");
    return 0;
}
