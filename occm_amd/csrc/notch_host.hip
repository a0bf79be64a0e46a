// Host-side notch-filter design for RawBoost (no GPU work): genNotchCoeffs, RawBoost.py:28-48,
// i.e. scipy.signal.firwin(c,[f1,f2],window='hamming',fs) band-stops convolved together, scaled
// by 10^(G/20) / max|H| on scipy.signal.freqz's 512-point grid.  The random draws (fc, bw, c, G)
// are made by the caller so the reference's np.random stream order is preserved.
#include "occ_common.h"
#include <cmath>
#include <vector>

namespace {
inline double sinc_pi(double x) {
    const double y = M_PI * (x == 0.0 ? 1.0e-20 : x);
    return std::sin(y) / y;
}
}  // namespace

extern "C" int occ_notch_coeffs_host(const double* fc, const double* bw, const int32_t* c, int64_t n_bands, double G, double fs,
                                     double* out_host, int32_t* ntaps_out_host, int64_t max_taps) {
    OCC_CHECK_ARG(fc && bw && c && out_host && ntaps_out_host && n_bands >= 1, "occ_notch_coeffs_host: bad argument");
    std::vector<double> b(1, 1.0);
    const double nyq = 0.5 * fs;
    for (int64_t i = 0; i < n_bands; ++i) {
        int ci = c[i];
        if (ci % 2 == 0) ci += 1;                            // RawBoost.py:35-36
        OCC_CHECK_ARG(ci >= 1, "occ_notch_coeffs_host: non-positive tap count");
        double f1 = fc[i] - bw[i] / 2, f2 = fc[i] + bw[i] / 2;
        if (f1 <= 0) f1 = 1.0 / 1000;                        // :39-42
        if (f2 >= fs / 2) f2 = fs / 2 - 1.0 / 1000;
        const double lo = f1 / nyq, hi = f2 / nyq;
        std::vector<double> h(ci);
        const double alpha = 0.5 * (ci - 1);
        double sum = 0.0;
        for (int n = 0; n < ci; ++n) {
            const double m = n - alpha;
            double v = lo * sinc_pi(lo * m);
            v = v + (1.0 * sinc_pi(1.0 * m) - hi * sinc_pi(hi * m));
            const double win = ci == 1 ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * M_PI * n / (ci - 1));
            h[n] = v * win;
            sum += h[n];
        }
        for (int n = 0; n < ci; ++n) h[n] /= sum;
        std::vector<double> nb(b.size() + ci - 1, 0.0);      // np.convolve(h, b)
        for (size_t p = 0; p < nb.size(); ++p) {
            double acc = 0.0;
            const size_t j0 = p >= b.size() - 1 ? p - (b.size() - 1) : 0;
            for (size_t j = j0; j < (size_t)ci && j <= p; ++j) acc += h[j] * b[p - j];
            nb[p] = acc;
        }
        b.swap(nb);
    }
    OCC_CHECK_ARG((int64_t)b.size() <= max_taps, "occ_notch_coeffs_host: %zu taps exceed max_taps %ld", b.size(), (long)max_taps);
    double hmax = 0.0;
    for (int k = 0; k < 512; ++k) {                          // freqz default grid, whole=False
        const double w = M_PI * k / 512.0;
        double re = 0.0, im = 0.0;
        for (size_t n = 0; n < b.size(); ++n) { re += b[n] * std::cos(w * n); im -= b[n] * std::sin(w * n); }
        const double mag = std::sqrt(re * re + im * im);
        if (mag > hmax) hmax = mag;
    }
    const double scale = std::pow(10.0, G / 20.0) / hmax;
    for (size_t n = 0; n < b.size(); ++n) out_host[n] = b[n] * scale;
    for (int64_t n = (int64_t)b.size(); n < max_taps; ++n) out_host[n] = 0.0;
    *ntaps_out_host = (int32_t)b.size();
    return OCC_OK;
}
