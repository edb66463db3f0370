// Host check of csrc/fft64_lds.h (compiled with g++ by tests/test_fft64_host.py): the stage code of the LDS transform, run
// thread by thread on an array that stands for the LDS image, against a direct DFT in long double; the inverse pass; and the
// bank-conflict count of every wave access under the two 16-byte rules of MI355X_MICROARCH.md.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft64_lds.h"

using namespace fft64;

static int total_conflicts(const Plan& pl, int threads) {
    int cost = 0;
    for (int st = 0; st < pl.nst; ++st) {
        const int nb = stage_count(pl, st);
        for (int w0 = 0; w0 < nb; w0 += 64)
            for (int m = 0; m < pl.radix[st]; ++m) {
                int e[64];
                for (int l = 0; l < 64; ++l) {
                    const int idx = w0 + l;
                    if (idx >= nb) { e[l] = -1; continue; }
                    int t;
                    e[l] = phys(pl, stage_base(pl, st, idx, &t)) ^ pl.pm[st][m];
                }
                cost += conflicts_b128(e, false) + conflicts_b128(e, true);
            }
    }
    (void)threads;
    return cost;
}

int main(int argc, char** argv) {
    int fails = 0;
    for (int ai = 1; ai < argc; ++ai) {
        const int need = std::atoi(argv[ai]);
        Plan pl = make_plan(need);
        const int N = pl.N;
        std::vector<double> twh = make_twiddles(pl);
        const f64c* tw = reinterpret_cast<const f64c*>(twh.data());
        std::vector<f64c> in(N), x(N);
        unsigned s = 12345u + (unsigned)N;
        for (int i = 0; i < N; ++i) {
            s = s * 1664525u + 1013904223u;
            const double a = (double)(s >> 8) / 16777216.0 - 0.5;
            s = s * 1664525u + 1013904223u;
            const double b = (double)(s >> 8) / 16777216.0 - 0.5;
            in[i] = mk(a, b);
        }
        // bijection of the image
        std::vector<int> hit(N, 0);
        bool bij = true;
        for (int p = 0; p < N; ++p) {
            const int q = phys(pl, p);
            if (q < 0 || q >= N || hit[q]++) bij = false;
        }
        for (int p = 0; p < N; ++p) x[phys(pl, p)] = in[p];
        const int TH = 256;
        for (int st = 0; st < pl.nst; ++st)
            for (int tid = 0; tid < TH; ++tid) stage_any<false>(x.data(), 0, 1, pl, st, tw, tid, TH);
        // direct DFT (long double) at all frequencies for small N, a sample for the large ones
        double worst = 0.0, scale = 0.0;
        const long double tau = 2.0L * 3.14159265358979323846264338327950288L;
        const int stride = N <= 2304 ? 1 : 37;
        std::vector<long double> ct(N), st_(N);
        for (int n = 0; n < N; ++n) {
            ct[n] = cosl(-tau * (long double)n / (long double)N);
            st_[n] = sinl(-tau * (long double)n / (long double)N);
        }
        for (int k = 0; k < N; k += stride) {
            long double re = 0, im = 0;
            for (int n = 0; n < N; ++n) {
                const int r = (int)(((long long)k * n) % N);
                const long double c = ct[r], sn = st_[r];
                re += in[n].x * c - in[n].y * sn;
                im += in[n].x * sn + in[n].y * c;
            }
            const f64c got = x[phys(pl, pos_of_freq(pl, k))];
            const double d = std::fabs((double)(got.x - re)) + std::fabs((double)(got.y - im));
            worst = d > worst ? d : worst;
            const double m = std::fabs((double)re) + std::fabs((double)im);
            scale = m > scale ? m : scale;
        }
        // backward pass on the conjugate -> N * conj(in)
        for (int p = 0; p < N; ++p) x[p] = cconj(x[p]);
        for (int st = pl.nst - 1; st >= 0; --st)
            for (int tid = 0; tid < TH; ++tid) stage_any<true>(x.data(), 0, 1, pl, st, tw, tid, TH);
        double back = 0.0;
        for (int p = 0; p < N; ++p) {
            const f64c got = x[phys(pl, p)];
            const double d = std::fabs(got.x / N - in[p].x) + std::fabs(-got.y / N - in[p].y);
            back = d > back ? d : back;
        }
        const int conf = total_conflicts(pl, TH);
        std::printf("need %d N %d R1 %d a %d stages", need, N, pl.R1, pl.a);
        for (int st = 0; st < pl.nst; ++st) std::printf(" %d", pl.radix[st]);
        std::printf(" | bijective %d fwd_err %.3g (scale %.3g) back_err %.3g conflicts %d masks %x %x %x %x\n", bij ? 1 : 0, worst, scale, back, conf,
                    pl.fmask[0], pl.fmask[1], pl.fmask[2], pl.fmask[3]);
        if (!bij || worst > 1e-12 * (scale > 1 ? scale : 1) || back > 1e-13) ++fails;
    }
    return fails ? 1 : 0;
}
