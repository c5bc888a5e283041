// grid_host.cpp -- per-element quadrature templates and functional specifications (host side).
//
// The molecular grid is never materialised on the host: for each ELEMENT the engine builds the
// atom-centred template (Treutler-Ahlrichs M4 radial mesh x NWChem-pruned Lebedev spheres, product
// weights 4 pi r^2 dr w_leb) once, uploads it, and the device composes fragment grids as
// atom position + template point, computing the Becke/Treutler partition weights in a kernel
// (kern_xc.hip).  The recipe follows the reference's CPU grid (src/methods/mqc_dft_grid.f90:145-257,
// mqc_dft_radial.f90:73-112, mqc_dft_prune.f90:44-133), NOT the unpruned cuEST-backend grid,
// because parity is against the CPU path.
#include "engine.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include "lebedev_tables.inc"

namespace mqc {

static const int PERIOD_LAST_Z[7] = {2, 10, 18, 36, 54, 86, 118};
static const int RAD_GRIDS[10][7] = {{10, 15, 20, 30, 35, 40, 50}, {30, 40, 50, 60, 65, 70, 75}, {40, 60, 65, 75, 80, 85, 90},
                                     {50, 75, 80, 90, 95, 100, 105}, {60, 90, 95, 105, 110, 115, 120}, {70, 105, 110, 120, 125, 130, 135},
                                     {80, 120, 125, 135, 140, 145, 150}, {90, 135, 140, 150, 155, 160, 165},
                                     {100, 150, 155, 165, 170, 175, 180}, {200, 200, 200, 200, 200, 200, 200}};
static const int ANG_POINTS[10][7] = {{50, 86, 110, 110, 110, 110, 110}, {110, 194, 194, 194, 194, 194, 194},
                                      {194, 302, 302, 302, 302, 302, 302}, {302, 302, 434, 434, 434, 434, 434},
                                      {434, 590, 590, 590, 590, 590, 590}, {590, 770, 770, 770, 770, 770, 770},
                                      {770, 974, 974, 974, 974, 974, 974}, {974, 1202, 1202, 1202, 1202, 1202, 1202},
                                      {1202, 1202, 1202, 1202, 1202, 1202, 1202}, {1454, 1454, 1454, 1454, 1454, 1454, 1454}};
// Z = 0 (ghost) .. 36: Treutler-Ahlrichs xi (JCP 102, 346) and Bragg-Slater radii in Angstrom
static const double TREUTLER_XI[37] = {1.0, 0.8, 0.9, 1.8, 1.4, 1.3, 1.1, 0.9, 0.9, 0.9, 0.9, 1.4, 1.3, 1.3, 1.2, 1.1, 1.0, 1.0, 1.0,
                                       1.5, 1.4, 1.3, 1.2, 1.2, 1.2, 1.2, 1.2, 1.2, 1.1, 1.1, 1.1, 1.1, 1.0, 0.9, 0.9, 0.9, 0.9};
static const double BRAGG_ANGSTROM[37] = {2.0, 0.35, 1.40, 1.45, 1.05, 0.85, 0.70, 0.65, 0.60, 0.50, 1.50, 1.80, 1.50, 1.25, 1.10, 1.00,
                                          1.00, 1.00, 1.80, 2.20, 1.80, 1.60, 1.40, 1.35, 1.40, 1.40, 1.40, 1.35, 1.35, 1.35, 1.35, 1.30,
                                          1.25, 1.15, 1.15, 1.15, 1.90};
static const int PRUNE_ORDERS[18] = {38, 50, 74, 86, 110, 146, 170, 194, 230, 266, 302, 350, 434, 590, 770, 974, 1202, 1454};
static const double ALPHAS[3][4] = {{0.25, 0.5, 1.0, 4.5}, {0.1667, 0.5, 0.9, 3.5}, {0.1, 0.4, 0.8, 2.5}};

double bragg_radius_bohr(int z)
{
    if (z < 0 || z > 36) return 1.0;
    return BRAGG_ANGSTROM[z] / 0.52917721092;    // the factor the reference's table was built with
}

static int element_period(int z)
{
    for (int i = 0; i < 7; ++i) if (z <= PERIOD_LAST_Z[i]) return i;
    return 6;
}

static int lebedev_index(int npts)
{
    for (int i = 0; i < LEBEDEV_NORDERS; ++i) if (LEBEDEV_ORDERS[i] == npts) return i;
    return -1;
}

// -> points relative to the nucleus [x,y,z]*n and product weights; returns false on an unsupported request
bool build_atom_template(int z, int level, int n_radial, int n_angular, std::vector<double>& xyz, std::vector<double>& w, std::string& err)
{
    if (z < 0 || z > 36) { err = "the XC grid tables cover elements up to Kr"; return false; }
    level = std::max(0, std::min(9, level));
    const int per = element_period(z);
    int nr = RAD_GRIDS[level][per], na = ANG_POINTS[level][per];
    if (n_radial > 0 && n_angular > 0) { nr = n_radial; na = n_angular; }
    else if ((n_radial > 0) != (n_angular > 0)) { err = "grid: n_radial and n_angular must be given together"; return false; }
    // radial mesh, ascending
    std::vector<double> r(nr), dr(nr);
    const double xi = TREUTLER_XI[z], step = M_PI / (nr + 1), scale = xi / std::log(2.0);
    for (int i = 1; i <= nr; ++i) {
        const double x = std::cos(i * step), s = std::sin(i * step);
        const double lt = std::log((1.0 - x) / 2.0), mt = std::pow(1.0 + x, 0.6);
        const int j = nr - i;
        r[j] = -scale * mt * lt;
        dr[j] = step * s * scale * mt * (-0.6 / (1.0 + x) * lt + 1.0 / (1.0 - x));
    }
    // NWChem pruning
    int zone[5];
    if (na < 50) { for (int k = 0; k < 5; ++k) zone[k] = na; }
    else if (na == 50) { zone[0] = 50; zone[1] = zone[2] = zone[3] = 74; zone[4] = 50; }
    else {
        int t = -1;
        for (int i = 0; i < 18; ++i) if (PRUNE_ORDERS[i] == na) t = i;
        if (t < 1) { err = "prune: target order is not a Lebedev order"; return false; }
        zone[0] = PRUNE_ORDERS[1]; zone[1] = PRUNE_ORDERS[3]; zone[2] = PRUNE_ORDERS[t - 1]; zone[3] = PRUNE_ORDERS[t]; zone[4] = PRUNE_ORDERS[t - 1];
    }
    const int cls = z <= 2 ? 0 : (z <= 10 ? 1 : 2);
    const double rb = bragg_radius_bohr(z) + 1e-200;
    xyz.clear(); w.clear();
    for (int i = 0; i < nr; ++i) {
        int order = na;
        if (na >= 50) {
            const double scaled = r[i] / rb;
            int zc = 0;
            for (int k = 0; k < 4; ++k) if (scaled > ALPHAS[cls][k]) ++zc;
            order = zone[zc];
        }
        const int li = lebedev_index(order);
        if (li < 0) { err = "no Lebedev table for " + std::to_string(order) + " points"; return false; }
        for (int k = LEBEDEV_OFFSETS[li]; k < LEBEDEV_OFFSETS[li + 1]; ++k) {
            xyz.push_back(r[i] * LEBEDEV_XYZW[k][0]); xyz.push_back(r[i] * LEBEDEV_XYZW[k][1]); xyz.push_back(r[i] * LEBEDEV_XYZW[k][2]);
            w.push_back(4.0 * M_PI * r[i] * r[i] * dr[i] * LEBEDEV_XYZW[k][3]);
        }
    }
    return true;
}

// functional name -> libxc-equivalent components (src/methods/mqc_xc_spec.f90:135-242)
bool parse_functional(const char* name_in, XcSpec& spec, std::string& err)
{
    std::string n(name_in);
    std::transform(n.begin(), n.end(), n.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    spec = XcSpec();
    auto add = [&spec](int id, double wgt) { spec.id[spec.ncomp] = id; spec.w[spec.ncomp] = wgt; spec.ncomp++; };
    if (n.empty()) { spec.exx = 1.0; return true; }                       // Hartree-Fock
    if (n == "svwn" || n == "lda" || n == "lsda") { add(XC_LDA_X, 1.0); add(XC_LDA_C_VWN, 1.0); spec.exx = 0.0; }
    else if (n == "pbe") { add(XC_GGA_X_PBE, 1.0); add(XC_GGA_C_PBE, 1.0); spec.exx = 0.0; spec.gga = 1; }
    else if (n == "blyp") { add(XC_GGA_X_B88, 1.0); add(XC_GGA_C_LYP, 1.0); spec.exx = 0.0; spec.gga = 1; }
    else if (n == "b3lyp") {   // libxc hyb_gga_xc_b3lyp: VWN-RPA flavour, 20 % exact exchange
        add(XC_LDA_X, 0.08); add(XC_GGA_X_B88, 0.72); add(XC_LDA_C_VWN_RPA, 0.19); add(XC_GGA_C_LYP, 0.81);
        spec.exx = 0.20; spec.gga = 1;
    }
    else if (n == "pbe0") { add(XC_GGA_X_PBE, 0.75); add(XC_GGA_C_PBE, 1.0); spec.exx = 0.25; spec.gga = 1; }
    else if (n == "tpss") { add(XC_MGGA_X_TPSS, 1.0); add(XC_MGGA_C_TPSS, 1.0); spec.exx = 0.0; spec.gga = 2; }     // mqc_xc_spec.f90:142-166
    else { err = "functional '" + n + "' is not available on the HIP backend (svwn, pbe, blyp, b3lyp, pbe0, tpss)"; return false; }
    return true;
}

}  // namespace mqc
