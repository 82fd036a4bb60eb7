/* cover_stats.cpp — DEV ONLY: how many fully covered Lee-Richards slices a cheap pre-pass would recognise.
 *   g++ -O2 -o /tmp/cover_stats tools/dev/cover_stats.cpp tools/synth.c -lm && /tmp/cover_stats globule 10000 2
 * For every slice that is neither buried by one neighbor nor open: is the circle already covered by the arcs of the
 * K neighbors with the largest caps on the atom's sphere (slice-independent choice), or by the K largest arcs of the
 * slice (per-slice choice)?  Not part of the product, not part of the oracle. */
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
struct Pair { int j; double beta, xd, yd, zd, d3sq, dij, costheta; };
static double acos_lb(double c)
{
    if (c > 0.9) return -1;
    const double k5 = c < 0 ? 0.075 : 0.17, c2 = c * c;
    return (1.5707963267948966 - 1e-9) - c - c * c2 * (1.0 / 6 + k5 * c2);
}
/* arcs in beta order (mid, half width): one running component, restart on a gap */
static bool covers_top_only(const std::vector<std::pair<double, double>> &ba)
{
    double ts = 0, te = -1e300;
    for (auto &x : ba) {
        if (x.second < 0) continue;
        const double inf = x.first - x.second, sup = x.first + x.second;
        if (inf > te) { ts = inf; te = sup; } else { ts = std::min(ts, inf); te = std::max(te, sup); }
    }
    return te - ts >= 2 * M_PI;
}
/* two components, give up on a third */
static bool covers_two(const std::vector<std::pair<double, double>> &ba)
{
    double ts = 0, te = -1e300, bs = 0, be = -1e300; int depth = 0;
    for (auto &x : ba) {
        if (x.second < 0) continue;
        const double inf = x.first - x.second, sup = x.first + x.second;
        if (inf > te) { if (depth == 2) return false; bs = ts; be = te; ts = inf; te = sup; ++depth; }
        else { ts = std::min(ts, inf); te = std::max(te, sup); if (depth == 2 && be >= ts) { ts = std::min(ts, bs); be = -1e300; depth = 1; } }
    }
    return depth == 1 && te - ts >= 2 * M_PI;
}
static bool covers(std::vector<std::pair<double, double>> as)
{
    std::vector<std::pair<double, double>> iv;
    for (auto &a : as) {
        double lo = a.first, hi = a.second;
        if (lo < 0) { iv.push_back({lo + 2 * M_PI, 2 * M_PI}); lo = 0; }
        if (hi > 2 * M_PI) { iv.push_back({0, hi - 2 * M_PI}); hi = 2 * M_PI; }
        iv.push_back({lo, hi});
    }
    std::sort(iv.begin(), iv.end());
    double sup = 0;
    for (auto &v : iv) { if (v.first > sup + 1e-9) return false; sup = std::max(sup, v.second); }
    return sup >= 2 * M_PI - 1e-9;
}
int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "globule";
    const int n = argc > 2 ? atoi(argv[2]) : 10000, nstruct = argc > 3 ? atoi(argv[3]) : 2, ns = 20;
    const double probe = 1.4;
    const int Ks[6] = {2, 3, 4, 6, 8, 12};
    double slices = 0, buried = 0, open_ = 0, full = 0, det_cap[6] = {0}, det_arc[6] = {0}, arcs_full = 0, arcs_all = 0, arcs_undet[6] = {0};
    double atoms = 0, atoms_zero = 0, atoms_det[6] = {0};
    const double taus[3] = {0.5, 0.6, 0.7}; const int Kt[3] = {8, 12, 16};
    double cull_n[3] = {0}, cull_cont[3] = {0}, cull_tot = 0;
    double det_top = 0, det_two = 0, det_two_exact = 0;
    double det_thr[3][3] = {{0}}, undet_thr[3][3] = {{0}}, ncont_thr[3][3] = {{0}}, maxarcs_thr[3][3] = {{0}};
    FILE *dump = fopen("/tmp/cover_items.txt", "w");
    for (int st = 0; st < nstruct; ++st) {
        std::vector<double> xyz(3 * n), rad(n), R(n);
        if (!strcmp(kind, "coil")) synth_coil(n, 1000 + st, xyz.data(), rad.data());
        else synth_globule(n, 500 + st, 2.6, xyz.data(), rad.data());
        double rmax = 0, lo[3] = {1e300, 1e300, 1e300};
        for (int i = 0; i < n; ++i) { R[i] = rad[i] + probe; rmax = std::max(rmax, R[i]); for (int k = 0; k < 3; ++k) lo[k] = std::min(lo[k], xyz[3 * i + k]); }
        const double d = 2 * rmax;
        std::map<long long, std::vector<int>> members;
        auto key = [&](int i, int dx, int dy, int dz) {
            long long c[3]; const int dd[3] = {dx, dy, dz};
            for (int k = 0; k < 3; ++k) c[k] = (long long)((xyz[3 * i + k] - lo[k]) / d) + 1 + dd[k];
            return c[0] + 4096 * (c[1] + 4096 * c[2]);
        };
        for (int i = 0; i < n; ++i) members[key(i, 0, 0, 0)].push_back(i);
        for (int i = 0; i < n; ++i) {
            std::vector<Pair> L;
            for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                auto it = members.find(key(i, dx, dy, dz));
                if (it == members.end()) continue;
                for (int j : it->second) {
                    if (j == i) continue;
                    const double xd = xyz[3 * j] - xyz[3 * i], yd = xyz[3 * j + 1] - xyz[3 * i + 1], zd = xyz[3 * j + 2] - xyz[3 * i + 2];
                    const double d3 = xd * xd + yd * yd + zd * zd, cut = R[i] + R[j];
                    if (d3 < cut * cut) {
                        Pair p; p.j = j; p.xd = xd; p.yd = yd; p.zd = zd; p.d3sq = d3; p.dij = sqrt(xd * xd + yd * yd);
                        p.beta = atan2(yd, xd) + M_PI; p.costheta = (R[i] * R[i] + d3 - R[j] * R[j]) / (2 * R[i] * sqrt(d3));
                        L.push_back(p);
                    }
                }
            }
            std::vector<int> idx(L.size());
            for (size_t k = 0; k < L.size(); ++k) idx[k] = (int)k;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return L[a].costheta < L[b].costheta; });
            std::vector<int> caprank(L.size());
            for (size_t k = 0; k < L.size(); ++k) caprank[idx[k]] = (int)k;
            { /* neighbor culling with bin-selected containers: WANT largest caps by 0.1 bins of cos(theta) from 0.2, KMAX slots in discovery order */
                for (int wi = 0; wi < 3; ++wi) {
                    const int WANT = wi == 0 ? 3 : (wi == 1 ? 4 : 6), KMAX = wi == 0 ? 4 : (wi == 1 ? 5 : 8);
                    int hist[8] = {0};
                    std::vector<int> bin(L.size());
                    for (size_t k = 0; k < L.size(); ++k) { int b = (int)(L[k].costheta * 10 - 2); b = b < 0 ? 0 : (b > 7 ? 7 : b); bin[k] = b; hist[b]++; }
                    int cum = 0, tb = 7;
                    for (int b = 0; b < 8; ++b) { cum += hist[b]; if (cum >= WANT && b < tb) tb = b; }
                    std::vector<int> cont;
                    for (size_t k = 0; k < L.size() && (int)cont.size() < KMAX; ++k) if (bin[k] <= tb) cont.push_back((int)k);
                    int nc = 0;
                    for (size_t j = 0; j < L.size(); ++j) {
                        bool cul = false;
                        const double thj = acos(std::max(-1.0, std::min(1.0, L[j].costheta)));
                        for (int k : cont) {
                            if (k == (int)j || !(L[k].costheta < L[j].costheta - 1e-4)) continue;
                            const double thk = acos(std::max(-1.0, std::min(1.0, L[k].costheta)));
                            const double cg = (L[j].xd * L[k].xd + L[j].yd * L[k].yd + L[j].zd * L[k].zd) / sqrt(L[j].d3sq * L[k].d3sq);
                            if (acos(std::max(-1.0, std::min(1.0, cg))) + thj <= thk - 1e-6) { cul = true; break; }
                        }
                        if (cul) ++nc;
                    }
                    cull_n[wi] += nc; cull_cont[wi] += cont.size();
                }
                cull_tot += L.size();
            }
            /* containers by threshold: the first K neighbors (discovery order) with cos(theta) < tau */
            std::vector<char> isc[3][3];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
                isc[a][b].assign(L.size(), 0);
                int cnt = 0;
                for (size_t k = 0; k < L.size() && cnt < Kt[b]; ++k) if (L[k].costheta < taus[a]) { isc[a][b][k] = 1; ++cnt; }
                ncont_thr[a][b] += cnt;
            }
            const double Ri = R[i], delta = 2 * Ri / ns;
            bool zero = true, detk[6] = {true, true, true, true, true, true};
            for (int s = 0; s < ns; ++s) {
                const double t = -Ri - 0.5 * delta + (s + 1) * delta, A = Ri * Ri - t * t;
                if (!(A > 0)) continue;
                slices += 1;
                const double Rip = sqrt(A);
                bool bur = false;
                std::vector<std::pair<double, double>> as;
                std::vector<double> al, cv, bv; std::vector<int> cr, li;
                for (size_t k = 0; k < L.size(); ++k) {
                    const Pair &p = L[k];
                    const double Kp = Ri * Ri - R[p.j] * R[p.j] + p.d3sq, c = (Kp - 2 * p.zd * t) / (2 * Rip * p.dij);
                    if (c >= 1) continue;
                    if (c <= -1) { bur = true; break; }
                    const double a = acos(c);
                    as.push_back({p.beta - a, p.beta + a}); al.push_back(a); cr.push_back(caprank[k]); li.push_back((int)k); cv.push_back(c); bv.push_back(p.beta);
                }
                if (bur) { buried += 1; fprintf(dump, "%d %d %d 1\n", st * n + i, s, 0); continue; }
                arcs_all += as.size();
                if (!covers(as)) { open_ += 1; zero = false; for (int q = 0; q < 6; ++q) detk[q] = false; fprintf(dump, "%d %d %d 2\n", st * n + i, s, (int)as.size()); continue; }
                full += 1; arcs_full += as.size();
                { /* tau 0.6, K 12, arcs in beta order (L is not sorted by beta here: sort) */
                    std::vector<std::pair<double, double>> ba, bx;
                    for (size_t k = 0; k < as.size(); ++k) if (isc[1][1][li[k]]) { ba.push_back({bv[k], acos_lb(cv[k])}); bx.push_back({bv[k], al[k]}); }
                    std::sort(ba.begin(), ba.end()); std::sort(bx.begin(), bx.end());
                    const bool dt = covers_top_only(ba);
                    fprintf(dump, "%d %d %d %d\n", st * n + i, s, (int)as.size(), dt ? 4 : 3);
                    if (dt) det_top += 1;
                    if (covers_two(ba)) det_two += 1;
                    if (covers_two(bx)) det_two_exact += 1;
                }
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) {
                    std::vector<std::pair<double, double>> sub;
                    for (size_t k = 0; k < as.size(); ++k) if (isc[a][b][li[k]]) sub.push_back(as[k]);
                    if (covers(sub)) det_thr[a][b] += 1; else undet_thr[a][b] += as.size();
                    maxarcs_thr[a][b] += sub.size();
                }
                std::vector<int> byarc(as.size());
                for (size_t k = 0; k < as.size(); ++k) byarc[k] = (int)k;
                std::sort(byarc.begin(), byarc.end(), [&](int a, int b) { return al[a] > al[b]; });
                for (int q = 0; q < 6; ++q) {
                    std::vector<std::pair<double, double>> sub, sub2;
                    for (size_t k = 0; k < as.size(); ++k) if (cr[k] < Ks[q]) sub.push_back(as[k]);
                    for (int k = 0; k < Ks[q] && k < (int)as.size(); ++k) sub2.push_back(as[byarc[k]]);
                    const bool c1 = covers(sub);
                    if (c1) det_cap[q] += 1; else { arcs_undet[q] += as.size(); detk[q] = false; }
                    if (covers(sub2)) det_arc[q] += 1;
                }
            }
            atoms += 1; if (zero) atoms_zero += 1;
            for (int q = 0; q < 6; ++q) if (detk[q]) atoms_det[q] += 1;
        }
    }
    fclose(dump);
    printf("%s: slices %.0f: buried %.3f open %.3f fully covered by several arcs %.3f; arcs in fully covered slices %.3f of all\n", kind, slices, buried / slices, open_ / slices, full / slices, arcs_full / arcs_all);
    for (int q = 0; q < 6; ++q)
        printf("K=%2d: covered slices recognised from the K largest caps %.3f (arcs left in unrecognised ones %.3f of all arcs), from the K largest arcs of the slice %.3f; atoms with zero area %.3f, recognised (buried or covered in every slice) %.3f\n",
               Ks[q], det_cap[q] / full, arcs_undet[q] / arcs_all, det_arc[q] / full, atoms_zero / atoms, atoms_det[q] / atoms);
    for (int wi = 0; wi < 3; ++wi) printf("culling, bin-selected containers (variant %d): %.2f containers/atom, culled %.3f of neighbors\n", wi, cull_cont[wi] / atoms, cull_n[wi] / cull_tot);
    printf("tau 0.6 K 12 with the lower bound of acos: one running component %.3f, two components %.3f (exact acos, two components %.3f)\n", det_top / full, det_two / full, det_two_exact / full);
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
        printf("tau %.1f K %2d: containers/atom %.1f, container arcs per covered slice %.1f, covered slices recognised %.3f (arcs left in unrecognised %.3f of all)\n", taus[a], Kt[b],
               ncont_thr[a][b] / atoms, maxarcs_thr[a][b] / full, det_thr[a][b] / full, undet_thr[a][b] / arcs_all);
    return 0;
}
