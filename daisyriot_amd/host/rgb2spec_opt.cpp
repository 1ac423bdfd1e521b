// rgb2spec_opt.cpp -- generator of the RGB -> spectrum coefficient table `color_tables/srgb.coeff` that the
// reference's Material reads at load time (visual studio/Material.cpp:11, rgb2spec.cpp:11-47) and that is
// absent from the reference tree (/root/reference/.MISSING_LARGE_BLOBS).  SURVEY.md 8(f)2.
//
//   rgb2spec_opt <resolution> <output.coeff>        (the reference's tables use resolution 64)
//
// File format (rgb2spec.cpp:16-43): "SPEC", uint32 res, float scale[res], float data[3][res][res][res][3].
// data[l][k][j][i] holds the three coefficients (c0,c1,c2) of the spectrum
//     s(lambda) = 1/2 + x / (2 sqrt(1 + x^2)),   x = c0 lambda^2 + c1 lambda + c2      (lambda in nm)
// (rgb2spec_eval_precise, rgb2spec.cpp:130-134) for the colour whose largest component l has the value scale[k]
// and whose other two components are i/(res-1) and j/(res-1) of it (rgb2spec_fetch, rgb2spec.cpp:78-121).
//
// Method: Jakob & Hanika, "A Low-Dimensional Function Space for Efficient Spectral Upsampling" (EG 2019): for
// every grid colour a Gauss-Newton fit of (c0,c1,c2), in CIELAB, so that the spectrum seen under the RGB space's
// white illuminant through the CIE 1931 observer gives back that colour; neighbouring grid cells warm-start each
// other outwards from brightness 1/5.  Restated from the paper; colour data used:
//   observer   -- the analytic CIE 1931 fit of Wyman, Sloan & Shirley (JCGT 2013), the same functions the
//                 reference's display path uses (visual studio/color.h:14-45), so that upsampling and display agree;
//   illuminant -- CIE standard illuminant D65, 10 nm table, linearly interpolated;
//   RGB space  -- sRGB / Rec.709 primaries with the matrices of visual studio/color.h:48-58.
// The original srgb.coeff was made from the tabulated 5 nm observer, so the numbers differ in the last digits of
// the spectra: parity with the missing blob is unpinned (it cannot be fetched); what the tests pin is the property
// the table exists for -- spectra in [0,1] that reproduce their RGB colour.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr double LAMBDA_MIN = 360.0, LAMBDA_MAX = 830.0;
constexpr int COARSE = 95;                          // 5 nm
constexpr int FINE = (COARSE - 1) * 3 + 1;          // Simpson 3/8 panels
constexpr double FD_EPS = 1e-4;

// CIE standard illuminant D65, relative spectral power, 300..830 nm in 10 nm steps
const double D65_10NM[54] = {
    0.0341, 3.2945, 20.236, 37.0535, 39.9488, 44.9117, 46.6383, 52.0891, 49.9755, 54.6482, 82.7549, 91.486, 93.4318,
    86.6823, 104.865, 117.008, 117.812, 114.861, 115.923, 108.811, 109.354, 107.802, 104.79, 107.689, 104.405, 104.046,
    100.0, 96.3342, 95.788, 88.6856, 90.0062, 89.5991, 87.6987, 83.2886, 83.6992, 80.0268, 80.2146, 82.2778, 78.2842,
    69.7213, 71.6091, 74.349, 61.604, 69.8856, 75.087, 63.5927, 46.4182, 66.8054, 63.3828, 64.304, 59.4519, 51.959,
    57.4406, 60.3125 };

double d65(double lambda) {
    double x = (lambda - 300.0) / 10.0;
    int i = std::max(0, std::min(52, (int)std::floor(x)));
    double f = x - i;
    return D65_10NM[i] * (1.0 - f) + D65_10NM[i + 1] * f;
}

// Wyman, Sloan, Shirley: multi-lobe piecewise Gaussian fit of the CIE 1931 colour-matching functions
void observer(double w, double xyz[3]) {
    auto lobe = [](double x, double mu, double s_lo, double s_hi) { double t = (x - mu) * (x < mu ? s_lo : s_hi); return std::exp(-0.5 * t * t); };
    xyz[0] = 0.362 * lobe(w, 442.0, 0.0624, 0.0374) + 1.056 * lobe(w, 599.8, 0.0264, 0.0323) - 0.065 * lobe(w, 501.1, 0.0490, 0.0382);
    xyz[1] = 0.821 * lobe(w, 568.8, 0.0213, 0.0247) + 0.286 * lobe(w, 530.9, 0.0613, 0.0322);
    xyz[2] = 1.217 * lobe(w, 437.0, 0.0845, 0.0278) + 0.681 * lobe(w, 459.0, 0.0385, 0.0725);
}

const double XYZ_TO_RGB[3][3] = { { 3.240479, -1.537150, -0.498535 }, { -0.969256, 1.875991, 0.041556 }, { 0.055648, -0.204043, 1.057311 } };
const double RGB_TO_XYZ[3][3] = { { 0.412453, 0.357580, 0.180423 }, { 0.212671, 0.715160, 0.072169 }, { 0.019334, 0.119193, 0.950227 } };

double lambda_tbl[FINE], rgb_tbl[3][FINE], white_xyz[3];

void init_tables() {
    const double h = (LAMBDA_MAX - LAMBDA_MIN) / (FINE - 1);
    std::memset(rgb_tbl, 0, sizeof rgb_tbl);
    white_xyz[0] = white_xyz[1] = white_xyz[2] = 0.0;
    for (int i = 0; i < FINE; i++) {
        const double lambda = LAMBDA_MIN + i * h;
        double xyz[3];
        observer(lambda, xyz);
        double w = 3.0 / 8.0 * h * d65(lambda);
        if (i != 0 && i != FINE - 1) w *= ((i - 1) % 3 == 2) ? 2.0 : 3.0;
        lambda_tbl[i] = lambda;
        for (int k = 0; k < 3; k++) {
            for (int j = 0; j < 3; j++) rgb_tbl[k][i] += XYZ_TO_RGB[k][j] * xyz[j] * w;
            white_xyz[k] += xyz[k] * w;
        }
    }
    // The analytic observer under the tabulated D65 does not integrate to exactly the white of the RGB matrices
    // (X,Y,Z = RGB_TO_XYZ * (1,1,1)); each observer channel is scaled so that it does: the perfect reflector then
    // is RGB (1,1,1), as in the original tool, whose observer, illuminant and matrices agree by construction.
    double gain[3];
    for (int k = 0; k < 3; k++) {
        const double target = RGB_TO_XYZ[k][0] + RGB_TO_XYZ[k][1] + RGB_TO_XYZ[k][2];
        gain[k] = target / white_xyz[k];
        white_xyz[k] = target;
    }
    std::memset(rgb_tbl, 0, sizeof rgb_tbl);
    for (int i = 0; i < FINE; i++) {
        double xyz[3];
        observer(lambda_tbl[i], xyz);
        double w = 3.0 / 8.0 * h * d65(lambda_tbl[i]);
        if (i != 0 && i != FINE - 1) w *= ((i - 1) % 3 == 2) ? 2.0 : 3.0;
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++) rgb_tbl[k][i] += XYZ_TO_RGB[k][j] * xyz[j] * gain[j] * w;
    }
}

double sigmoid(double x) { return 0.5 * x / std::sqrt(1.0 + x * x) + 0.5; }
double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

void to_lab(double p[3]) {
    double xyz[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) xyz[k] += p[j] * RGB_TO_XYZ[k][j];
    auto f = [](double t) { const double d = 6.0 / 29.0; return t > d * d * d ? std::cbrt(t) : t / (3.0 * d * d) + 4.0 / 29.0; };
    const double fx = f(xyz[0] / white_xyz[0]), fy = f(xyz[1] / white_xyz[1]), fz = f(xyz[2] / white_xyz[2]);
    p[0] = 116.0 * fy - 16.0;
    p[1] = 500.0 * (fx - fy);
    p[2] = 200.0 * (fy - fz);
}

void spectrum_rgb(const double c[3], double out[3]) {
    out[0] = out[1] = out[2] = 0.0;
    for (int i = 0; i < FINE; i++) {
        const double l = (lambda_tbl[i] - LAMBDA_MIN) / (LAMBDA_MAX - LAMBDA_MIN);      // fit on [0,1]
        const double s = sigmoid((c[0] * l + c[1]) * l + c[2]);
        for (int j = 0; j < 3; j++) out[j] += rgb_tbl[j][i] * s;
    }
}

void residual(const double c[3], const double rgb[3], double r[3]) {
    double out[3], want[3] = { rgb[0], rgb[1], rgb[2] };
    spectrum_rgb(c, out);
    to_lab(out);
    to_lab(want);
    for (int j = 0; j < 3; j++) r[j] = want[j] - out[j];
}

// solve the 3x3 system J x = r by Gaussian elimination with partial pivoting; false if singular
bool solve3(double J[3][3], const double r[3], double x[3]) {
    double a[3][4];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) a[i][j] = J[i][j]; a[i][3] = r[i]; }
    for (int col = 0; col < 3; col++) {
        int piv = col;
        for (int i = col + 1; i < 3; i++) if (std::fabs(a[i][col]) > std::fabs(a[piv][col])) piv = i;
        if (std::fabs(a[piv][col]) < 1e-15) return false;
        if (piv != col) for (int j = 0; j < 4; j++) std::swap(a[piv][j], a[col][j]);
        for (int i = col + 1; i < 3; i++) {
            const double f = a[i][col] / a[col][col];
            for (int j = col; j < 4; j++) a[i][j] -= f * a[col][j];
        }
    }
    for (int i = 2; i >= 0; i--) {
        double s = a[i][3];
        for (int j = i + 1; j < 3; j++) s -= a[i][j] * x[j];
        x[i] = s / a[i][i];
    }
    return true;
}

double norm2(const double r[3]) { return r[0] * r[0] + r[1] * r[1] + r[2] * r[2]; }

// Gauss-Newton on the CIELAB residual (central-difference Jacobian), the paper's iteration, with one addition:
// a step is halved (up to 6 times) while it does not reduce the residual, which keeps fits at the gamut
// boundary from running into the coefficient clamp and staying there.  Returns the final squared residual.
double gauss_newton(const double rgb[3], double c[3], int iterations = 15) {
    double r[3];
    residual(c, rgb, r);
    double rr = norm2(r);
    for (int it = 0; it < iterations && rr >= 1e-6; it++) {
        double J[3][3];
        for (int i = 0; i < 3; i++) {
            double t[3] = { c[0], c[1], c[2] }, r0[3], r1[3];
            t[i] = c[i] - FD_EPS; residual(t, rgb, r0);
            t[i] = c[i] + FD_EPS; residual(t, rgb, r1);
            for (int j = 0; j < 3; j++) J[j][i] = (r1[j] - r0[j]) / (2 * FD_EPS);
        }
        double x[3];
        if (!solve3(J, r, x)) break;
        double step = 1.0, best_rr = rr, best_c[3] = { c[0], c[1], c[2] }, best_r[3] = { r[0], r[1], r[2] };
        for (int half = 0; half < 7; half++, step *= 0.5) {
            double t[3], tr[3], mx = 0;
            for (int j = 0; j < 3; j++) { t[j] = c[j] - step * x[j]; mx = std::max(mx, std::fabs(t[j])); }
            if (mx > 200.0) for (int j = 0; j < 3; j++) t[j] *= 200.0 / mx;   // keeps the sigmoid from saturating for good
            residual(t, rgb, tr);
            const double trr = norm2(tr);
            if (trr < best_rr) {
                best_rr = trr;
                for (int j = 0; j < 3; j++) { best_c[j] = t[j]; best_r[j] = tr[j]; }
                break;
            }
        }
        if (!(best_rr < rr)) break;                 // no reducing step: a (local) minimum
        rr = best_rr;
        for (int j = 0; j < 3; j++) { c[j] = best_c[j]; r[j] = best_r[j]; }
    }
    return rr;
}

// the fit of one colour from the warm start `c`; when that does not converge, from a flat spectrum of the colour's
// mean and from zero as well, keeping the best
double fit_colour(const double rgb[3], double c[3]) {
    const double TOL = 1e-4;                         // squared CIELAB distance
    double best[3] = { c[0], c[1], c[2] };
    double best_rr = gauss_newton(rgb, best, 15);
    if (best_rr > TOL) {
        const double m = std::min(std::max((rgb[0] + rgb[1] + rgb[2]) / 3.0, 1e-4), 1.0 - 1e-4), u = 2.0 * m - 1.0;
        double starts[2][3] = { { 0.0, 0.0, u / std::sqrt(1.0 - u * u) }, { 0.0, 0.0, 0.0 } };
        for (auto& st : starts) {
            const double rr = gauss_newton(rgb, st, 40);
            if (rr < best_rr) { best_rr = rr; best[0] = st[0]; best[1] = st[1]; best[2] = st[2]; }
            if (best_rr <= TOL) break;
        }
    }
    c[0] = best[0]; c[1] = best[1]; c[2] = best[2];
    return best_rr;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <resolution> <output.coeff>\n", argv[0]);
        return 1;
    }
    const int res = std::atoi(argv[1]);
    if (res < 2 || res > 256) { std::fprintf(stderr, "resolution must be in [2,256]\n"); return 1; }
    init_tables();
    std::vector<float> scale((size_t)res), data((size_t)3 * res * res * res * 3);
    for (int k = 0; k < res; k++) scale[(size_t)k] = (float)smoothstep(smoothstep((double)k / (res - 1)));
    for (int l = 0; l < 3; l++) {
#pragma omp parallel for schedule(dynamic)
        for (int j = 0; j < res; j++) {
            const double y = (double)j / (res - 1);
            for (int i = 0; i < res; i++) {
                const double x = (double)i / (res - 1);
                auto fit = [&](int k, double c[3]) {
                    const double b = scale[(size_t)k];
                    double rgb[3];
                    rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                    fit_colour(rgb, c);
                    // coefficients of lambda in nm instead of (lambda - 360) / 470
                    const double c0 = LAMBDA_MIN, c1 = 1.0 / (LAMBDA_MAX - LAMBDA_MIN);
                    const double A = c[0], B = c[1], C = c[2];
                    const size_t idx = ((((size_t)l * res + k) * res + j) * res + i) * 3;
                    data[idx + 0] = (float)(A * c1 * c1);
                    data[idx + 1] = (float)(B * c1 - 2 * A * c0 * c1 * c1);
                    data[idx + 2] = (float)(C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
                };
                // from brightness scale[res/5] upwards, then downwards; every cell starts from its neighbour's fit
                const int start = res / 5;
                double c[3] = { 0, 0, 0 };
                for (int k = start; k < res; k++) fit(k, c);
                c[0] = c[1] = c[2] = 0.0;
                for (int k = start; k >= 0; k--) fit(k, c);
            }
        }
    }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    const uint32_t r32 = (uint32_t)res;
    std::fwrite("SPEC", 4, 1, f);
    std::fwrite(&r32, sizeof r32, 1, f);
    std::fwrite(scale.data(), sizeof(float), scale.size(), f);
    std::fwrite(data.data(), sizeof(float), data.size(), f);
    std::fclose(f);
    std::printf("wrote %s: resolution %d, %zu coefficients\n", argv[2], res, data.size());
    return 0;
}
