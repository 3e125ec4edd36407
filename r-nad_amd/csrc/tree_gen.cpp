// tree_gen.cpp -- native (host) tree generator and zero-sum matrix-game solver.
//
// Replaces, for trees too large for the Python recursion: environment/tree.py:164-366 (Tree._init_child,
// _transition_probs, _solve, generate).  Citations are baskuit/R-NaD file:line.  The reference solves each state's
// matrix game with pygambit 16.0.2 (requirements.txt:3; `enummixed_solve`, tree.py:205-223), which is third-party, not
// vendored and not installable here; its published algorithm (enumeration of extreme equilibria) is restated for the
// zero-sum case via Shapley-Snow kernels, in the same sub-matrix order as tests/golden/_pygambit_stub.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <atomic>
#include <vector>

#include "rnad_hip.h"

namespace rnad {
void set_error(const char *fmt, ...);
}

namespace {

constexpr int MAXA = RNAD_MAX_ACTIONS;

// Solve B z = 1 (k x k) by Gaussian elimination with partial pivoting; returns false if |det| < 1e-12.
bool solve_ones(const double *Bm, int k, bool transpose, double *z) {
    double a[MAXA][MAXA + 1];
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) a[i][j] = transpose ? Bm[j * MAXA + i] : Bm[i * MAXA + j];
        a[i][k] = 1.0;
    }
    double det = 1.0;
    for (int col = 0; col < k; ++col) {
        int piv = col;
        for (int r = col + 1; r < k; ++r)
            if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
        if (a[piv][col] == 0.0) return false;
        if (piv != col) {
            for (int j = 0; j <= k; ++j) std::swap(a[piv][j], a[col][j]);
            det = -det;
        }
        det *= a[col][col];
        for (int r = col + 1; r < k; ++r) {
            const double f = a[r][col] / a[col][col];
            for (int j = col; j <= k; ++j) a[r][j] -= f * a[col][j];
        }
    }
    if (std::fabs(det) < 1e-12) return false;
    for (int i = k - 1; i >= 0; --i) {
        double s = a[i][k];
        for (int j = i + 1; j < k; ++j) s -= a[i][j] * z[j];
        z[i] = s / a[i][i];
    }
    return true;
}

struct Strat {
    double p[MAXA];
};

void add_unique(std::vector<Strat> &lst, const Strat &s, int n) {
    for (const Strat &w : lst) {
        double d = 0.0;
        for (int i = 0; i < n; ++i) d = std::max(d, std::fabs(w.p[i] - s.p[i]));
        if (d < 1e-7) return;
    }
    lst.push_back(s);
}

// next k-combination of {0..n-1} in lexicographic order (itertools.combinations order)
bool next_comb(int *c, int k, int n) {
    int i = k - 1;
    while (i >= 0 && c[i] == n - k + i) --i;
    if (i < 0) return false;
    ++c[i];
    for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
    return true;
}

// All extreme optimal strategies of the zero-sum game M (row maximises): Shapley-Snow kernels of M + shift.
void extreme_strategies(const double *M, int ra, int ca, std::vector<Strat> &xs, std::vector<Strat> &ys) {
    const double eps = 1e-9;
    double mn = M[0];
    for (int i = 0; i < ra; ++i)
        for (int j = 0; j < ca; ++j) mn = std::min(mn, M[i * MAXA + j]);
    double Bm[MAXA * MAXA];
    for (int i = 0; i < ra; ++i)
        for (int j = 0; j < ca; ++j) Bm[i * MAXA + j] = M[i * MAXA + j] + (1.0 - mn);  // every entry >= 1 => value > 0
    for (int k = 1; k <= std::min(ra, ca); ++k) {
        int rows[MAXA], cols[MAXA];
        for (int i = 0; i < k; ++i) rows[i] = i;
        do {
            for (int i = 0; i < k; ++i) cols[i] = i;
            do {
                double sub[MAXA * MAXA], yk[MAXA], xk[MAXA];
                for (int i = 0; i < k; ++i)
                    for (int j = 0; j < k; ++j) sub[i * MAXA + j] = Bm[rows[i] * MAXA + cols[j]];
                if (!solve_ones(sub, k, false, yk) || !solve_ones(sub, k, true, xk)) continue;
                double sy = 0.0, sx = 0.0, miny = yk[0], minx = xk[0];
                for (int i = 0; i < k; ++i) {
                    sy += yk[i]; sx += xk[i];
                    miny = std::min(miny, yk[i]); minx = std::min(minx, xk[i]);
                }
                if (miny < -eps || minx < -eps || sy <= eps || sx <= eps) continue;
                const double v = 1.0 / sy;
                Strat x{}, y{};
                for (int i = 0; i < k; ++i) {
                    x.p[rows[i]] = xk[i] / sx;
                    y.p[cols[i]] = yk[i] / sy;
                }
                bool ok = true;
                for (int j = 0; j < ca && ok; ++j) {  // x'B >= v
                    double s = 0.0;
                    for (int i = 0; i < ra; ++i) s += x.p[i] * Bm[i * MAXA + j];
                    ok = s >= v - 1e-7;
                }
                for (int i = 0; i < ra && ok; ++i) {  // B y <= v
                    double s = 0.0;
                    for (int j = 0; j < ca; ++j) s += Bm[i * MAXA + j] * y.p[j];
                    ok = s <= v + 1e-7;
                }
                if (!ok) continue;
                for (int i = 0; i < ra; ++i) {
                    if (std::fabs(x.p[i]) < 1e-12) x.p[i] = 0.0;
                    if (std::fabs(x.p[i] - 1.0) < 1e-12) x.p[i] = 1.0;
                }
                for (int j = 0; j < ca; ++j) {
                    if (std::fabs(y.p[j]) < 1e-12) y.p[j] = 0.0;
                    if (std::fabs(y.p[j] - 1.0) < 1e-12) y.p[j] = 1.0;
                }
                add_unique(xs, x, ra);
                add_unique(ys, y, ca);
            } while (next_comb(cols, k, ca));
        } while (next_comb(rows, k, ra));
        // A saddle point gives a (pure, pure) pair, the lowest purity key there is (tree.py:227-231), and the stable sort keeps
        // the FIRST such pair in enumeration order -- which is made of the first pure x and the first pure y found at k = 1.
        // Larger supports cannot change the selection any more, so the enumeration stops here.
        if (k == 1 && !xs.empty() && !ys.empty()) return;
    }
}

// tree.py:199-234: solutions = all (x, y) pairs; stable sort by purity score; take the first.  Returns false if none.
bool solve_matrix(const float *M, int ra, int ca, int max_actions, float *solution, float *value) {
    double Md[MAXA * MAXA];
    for (int i = 0; i < ra; ++i)
        for (int j = 0; j < ca; ++j) Md[i * MAXA + j] = (double)M[i * ca + j];
    std::vector<Strat> xs, ys;
    extreme_strategies(Md, ra, ca, xs, ys);
    if (xs.empty() || ys.empty()) return false;
    int best_score = 1, bx = 0, by = 0;
    for (size_t i = 0; i < xs.size(); ++i)
        for (size_t j = 0; j < ys.size(); ++j) {
            bool px = false, py = false;  // `1 in solution[:max_actions]`, `1 in solution[max_actions:]` (tree.py:227-229)
            for (int a = 0; a < ra; ++a) px |= xs[i].p[a] == 1.0;
            for (int a = 0; a < ca; ++a) py |= ys[j].p[a] == 1.0;
            const int score = -(int)px - (int)py;
            // list.sort(key=purity_score) is stable and ascending and the key is MINUS the number of pure sides, so the
            // first pair with the most pure sides ends up in front (tree.py:227-231; the comment there says otherwise).
            if (best_score == 1 || score < best_score) {
                best_score = score; bx = (int)i; by = (int)j;
            }
        }
    for (int a = 0; a < 2 * max_actions; ++a) solution[a] = 0.0f;
    for (int a = 0; a < ra; ++a) solution[a] = (float)xs[bx].p[a];
    for (int a = 0; a < ca; ++a) solution[max_actions + a] = (float)ys[by].p[a];
    if (value) {  // root_value = (p1 @ M) @ p2 in fp32 (tree.py:298-300)
        float acc = 0.0f;  // the BLAS behind torch.matmul accumulates these tiny dot products with fused multiply-adds
        for (int j = 0; j < ca; ++j) {
            float col = 0.0f;
            for (int i = 0; i < ra; ++i) col = std::fmaf(solution[i], M[i * ca + j], col);
            acc = std::fmaf(col, solution[max_actions + j], acc);
        }
        *value = acc;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------- random numbers
struct Rng {
    uint64_t s;
    uint64_t next() {  // splitmix64
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return ((next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0, 1)
    uint64_t below(uint64_t n) { return next() % n; }
    double normal() {
        const double u1 = uniform(), u2 = uniform();
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    double gamma(double a) {  // Marsaglia-Tsang, boosted for a < 1
        if (a < 1.0) return gamma(a + 1.0) * std::pow(uniform(), 1.0 / a);
        const double d = a - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
        for (;;) {
            double x, v;
            do {
                x = normal();
                v = 1.0 + c * x;
            } while (v <= 0.0);
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * x * x * x * x || std::log(u) < 0.5 * x * x + d * (1.0 - v + std::log(v))) return d * v;
        }
    }
};

struct Gen {
    int A, C;
    float thr;
    const float *tv;
    int ntv, prune_num, prune_den;
    Rng rng;
    int64_t capacity, count = 0;
    int64_t *index;
    float *value, *chance, *ev, *legal, *root_value, *solution;
    bool failed = false;

    bool writing() const { return index != nullptr; }

    // tree.py:182-197 for one (row, col): Dirichlet(1/C) -> zero entries < threshold -> L1 renormalise (fp32).
    void chance_profile(float *p) {
        if (C == 1) {  // Dirichlet over one outcome
            p[0] = 1.0f;
            return;
        }
        double g[RNAD_MAX_TRANSITIONS], s = 0.0;
        for (int t = 0; t < C; ++t) s += g[t] = rng.gamma(1.0 / C);
        float f[RNAD_MAX_TRANSITIONS], fs = 0.0f;
        for (int t = 0; t < C; ++t) {
            f[t] = (float)(g[t] / s);
            f[t] = f[t] - (f[t] < thr ? f[t] : 0.0f);
            fs += std::fabs(f[t]);
        }
        const float d = fs > 1e-12f ? fs : 1e-12f;
        for (int t = 0; t < C; ++t) p[t] = f[t] / d;
    }

    // Structure pass for one state (DFS pre-order id, root = 1): chance profiles, pruning, terminal payoffs, child ids --
    // everything that consumes random numbers, in the reference's order.  Matrix games are NOT solved here: a state's payoff
    // matrix needs its children's equilibrium payoffs, so the solves run afterwards, bottom-up by level and in parallel
    // (solve_levels).  The counting pass (no output buffers) stops at the structure.
    void build(int depth_bound, int level, int64_t parent_slot) {
        const int AA = A * A;
        const int64_t id = ++count;  // pre-order: the parent takes its id before its children (tree.py:311-330)
        if (writing() && id >= capacity) {
            failed = true;
            return;
        }
        std::vector<float> ch((size_t)C * AA);
        for (int rc = 0; rc < AA; ++rc) {  // the child-less ctor draws the whole [A, A, C] profile first (tree.py:134-136)
            float p[RNAD_MAX_TRANSITIONS];
            chance_profile(p);
            for (int t = 0; t < C; ++t) ch[(size_t)t * AA + rc] = p[t];
        }
        if (writing()) {
            for (int k = 0; k < C * AA; ++k) {
                index[id * C * AA + k] = 0;
                value[id * C * AA + k] = 0.0f;
                chance[id * C * AA + k] = ch[k];
            }
            for (int k = 0; k < AA; ++k) legal[id * AA + k] = 1.0f;
            level_of[(size_t)id] = level;
            parent_of[(size_t)id] = parent_slot;
            max_level = std::max(max_level, level);
        }
        for (int r = 0; r < A; ++r)
            for (int c = 0; c < A; ++c) {
                const int rc = r * A + c;
                for (int t = 0; t < C; ++t) {  // tree.py:253-277
                    if (!(ch[(size_t)t * AA + rc] > 0.0f)) continue;
                    int child_depth = depth_bound - 1;
                    if (prune_den > 0 && (int)rng.below((uint64_t)prune_den) < prune_num) child_depth -= 2;  // main.py:37
                    child_depth = std::max(0, child_depth);
                    const int64_t slot = id * C * AA + (int64_t)t * AA + rc;
                    if (child_depth > 0) {
                        if (writing()) index[slot] = count + 1;  // the id the child is about to take
                        build(child_depth, level + 1, slot);
                        if (failed) return;
                    } else {
                        const float payoff = tv[rng.below((uint64_t)ntv)];  // random.choice(terminal_values) (tree.py:273-275)
                        if (writing()) value[slot] = payoff;
                    }
                }
            }
    }

    // tree.py:280-300 for one state whose children are done: expected_value = sum_t value * chance, the matrix game, and the
    // state's own payoff handed up into its parent's value slot.
    bool solve_state(int64_t id) {
        const int AA = A * A;
        float evm[MAXA * MAXA];
        for (int rc = 0; rc < AA; ++rc) {
            float e = 0.0f;
            for (int t = 0; t < C; ++t) e += value[id * C * AA + (int64_t)t * AA + rc] * chance[id * C * AA + (int64_t)t * AA + rc];
            evm[rc] = e;
            ev[id * AA + rc] = e;
        }
        float sol[2 * MAXA], rv = 0.0f;
        if (!solve_matrix(evm, A, A, A, sol, &rv)) return false;
        root_value[id] = rv;
        for (int k = 0; k < 2 * A; ++k) solution[id * 2 * A + k] = sol[k];
        if (parent_of[(size_t)id] >= 0) value[parent_of[(size_t)id]] = rv;
        return true;
    }

    // Deepest level first; the states of one level are independent (each writes its own rows and one slot of its parent).
    void solve_levels() {
        std::vector<std::vector<int64_t>> by_level((size_t)max_level + 1);
        for (int64_t id = 1; id <= count; ++id) by_level[(size_t)level_of[(size_t)id]].push_back(id);
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned max_threads = std::max(1u, std::min(hw ? hw : 1u, 32u));
        std::atomic<bool> bad{false};
        for (int lv = max_level; lv >= 0; --lv) {
            const std::vector<int64_t> &ids = by_level[(size_t)lv];
            const size_t n = ids.size();
            const unsigned nt = (unsigned)std::min<size_t>(max_threads, (n + 255) / 256);  // >= 256 states per thread
            auto work = [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i)
                    if (!solve_state(ids[i])) bad = true;
            };
            if (nt <= 1) {
                work(0, n);
            } else {
                std::vector<std::thread> pool;
                for (unsigned w = 0; w < nt; ++w) pool.emplace_back(work, n * w / nt, n * (w + 1) / nt);
                for (std::thread &t : pool) t.join();
            }
        }
        if (bad) failed = true;
    }

    std::vector<int> level_of;        // [capacity] depth below the root
    std::vector<int64_t> parent_of;   // [capacity] flat index of the parent's value slot, -1 for the root
    int max_level = 0;
};

}  // namespace

extern "C" int rnad_solve_matrix(const float *M, int ra, int ca, int max_actions, float *solution, float *value) {
    if (!M || !solution || ra < 1 || ca < 1 || ra > max_actions || ca > max_actions || max_actions > MAXA) {
        rnad::set_error("rnad_solve_matrix: bad arguments (ra=%d ca=%d max_actions=%d)", ra, ca, max_actions);
        return 2;
    }
    if (!solve_matrix(M, ra, ca, max_actions, solution, value)) {
        rnad::set_error("rnad_solve_matrix: no equilibrium found");  // reference raises here too (tree.py:287-290)
        return 3;
    }
    return 0;
}

extern "C" int64_t rnad_tree_generate(int A, int C, int depth_bound, float transition_threshold, const float *terminal_values,
                                      int n_terminal_values, int prune_num, int prune_den, uint64_t seed, int64_t capacity,
                                      int64_t *index, float *value, float *chance, float *expected_value, float *legal,
                                      float *root_value, float *solution) {
    if (A < 1 || A > MAXA || C < 1 || C > RNAD_MAX_TRANSITIONS || depth_bound < 1 || !terminal_values || n_terminal_values < 1) {
        rnad::set_error("rnad_tree_generate: bad arguments (A=%d C=%d depth_bound=%d)", A, C, depth_bound);
        return -1;
    }
    if (C > 1 && !(transition_threshold < 1.0f / C)) {
        // a threshold >= 1/C can zero a whole profile; torch.multinomial then raises in the reference
        rnad::set_error("rnad_tree_generate: transition_threshold %g must be < 1/C", (double)transition_threshold);
        return -1;
    }
    const bool writing = index != nullptr;
    if (writing && !(value && chance && expected_value && legal && root_value && solution)) {
        rnad::set_error("rnad_tree_generate: some output buffers are null");
        return -1;
    }
    Gen g{A, C, transition_threshold, terminal_values, n_terminal_values, prune_num, prune_den, Rng{seed}, capacity};
    g.index = index; g.value = value; g.chance = chance; g.ev = expected_value; g.legal = legal;
    g.root_value = root_value; g.solution = solution;
    const int AA = A * A;
    if (writing) {
        if (capacity < 2) {
            rnad::set_error("rnad_tree_generate: capacity too small");
            return -1;
        }
        // absorbing state 0 (tree.py:338-349): one legal joint action, chance[0,0,0,0] = 1, everything else zero
        memset(index, 0, sizeof(int64_t) * C * AA);
        memset(value, 0, sizeof(float) * C * AA);
        memset(chance, 0, sizeof(float) * C * AA);
        memset(expected_value, 0, sizeof(float) * AA);
        memset(legal, 0, sizeof(float) * AA);
        memset(solution, 0, sizeof(float) * 2 * A);
        chance[0] = 1.0f;
        legal[0] = 1.0f;
        root_value[0] = 0.0f;
    }
    if (writing) {
        g.level_of.assign((size_t)capacity, 0);
        g.parent_of.assign((size_t)capacity, -1);
    }
    g.build(depth_bound, 0, -1);
    if (writing && !g.failed) g.solve_levels();
    if (g.failed) {
        rnad::set_error(writing ? "rnad_tree_generate: capacity %lld too small or a matrix was not solved"
                                : "rnad_tree_generate: a matrix game was not solved",
                        (long long)capacity);
        return -1;
    }
    return g.count + 1;  // states incl. the absorbing one
}
