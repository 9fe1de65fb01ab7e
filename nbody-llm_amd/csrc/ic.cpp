// ic.cpp -- synthetic initial conditions (host side, f64 arithmetic, rounded to f32 records).
//
// nbody_ic_disc restates the reference's only IC generator, the self-gravitating disc of
// src/main.rs:52-89 (1 unit-mass star + n disc bodies), with the repository's own PRNG in place
// of rand::random (the reference seeds from the OS, so no sequence exists to reproduce).
// nbody_ic_plummer is the Plummer sphere BASELINE.json's configs are quoted on (Aarseth, Henon &
// Wielen 1974 sampling), which the reference does not have.
#include "../../include/nbody_hip.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Xoshiro256ss {  // xoshiro256** seeded by splitmix64 (public-domain algorithms, Blackman & Vigna)
    uint64_t s[4];
    explicit Xoshiro256ss(uint64_t seed) {
        uint64_t z = seed;
        for (int i = 0; i < 4; ++i) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t x = z;
            x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
            x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
            s[i] = x ^ (x >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0, 1)
};

// records are PointParticle<f32,3> (40 B) or, `wide`, PointParticle<f64,3> (80 B, the values unrounded)
void put(void* aos, size_t k, size_t stride, const double pos[3], const double vel[3], double m, bool wide) {
    if (wide) {
        double rec[10] = {pos[0], pos[1], pos[2], vel[0], vel[1], vel[2], 0.0, 0.0, 0.0, m};
        std::memcpy(static_cast<char*>(aos) + k * stride, rec, sizeof(rec));
        return;
    }
    float rec[10] = {float(pos[0]), float(pos[1]), float(pos[2]), float(vel[0]), float(vel[1]), float(vel[2]),
                     0.f, 0.f, 0.f, float(m)};
    std::memcpy(static_cast<char*>(aos) + k * stride, rec, sizeof(rec));
}

}  // namespace

static int ic_plummer(void* aos, size_t n, size_t stride, uint64_t seed, bool wide) {
    if (!aos || stride < (wide ? 80u : 40u)) return NBODY_ERR_INVALID;
    if (n == 0) return NBODY_OK;
    const double kPi = 3.14159265358979323846;
    const double len = 3.0 * kPi / 16.0;  // Henon units: G = M = 1, E = -1/4
    const double vsc = std::sqrt(1.0 / len);
    const double r_max = 10.0 / len;      // truncate at 10 Henon length units
    Xoshiro256ss rng(seed);
    std::vector<double> P(3 * n), V(3 * n);
    double cp[3] = {0, 0, 0}, cv[3] = {0, 0, 0};
    for (size_t k = 0; k < n; ++k) {
        double r;
        do {
            double x1;
            do { x1 = rng.uniform(); } while (x1 <= 0.0);
            r = 1.0 / std::sqrt(std::pow(x1, -2.0 / 3.0) - 1.0);
        } while (!(r <= r_max));
        double x2 = rng.uniform(), x3 = rng.uniform();
        double z = (1.0 - 2.0 * x2) * r;
        double rho = std::sqrt(std::fmax(0.0, r * r - z * z));
        double x = rho * std::cos(2.0 * kPi * x3), y = rho * std::sin(2.0 * kPi * x3);
        double q, gq;
        do {
            q = rng.uniform();
            gq = 0.1 * rng.uniform();
        } while (gq >= q * q * std::pow(1.0 - q * q, 3.5));
        double v = q * std::sqrt(2.0) * std::pow(1.0 + r * r, -0.25);
        double x6 = rng.uniform(), x7 = rng.uniform();
        double vz = (1.0 - 2.0 * x6) * v;
        double vr = std::sqrt(std::fmax(0.0, v * v - vz * vz));
        double vx = vr * std::cos(2.0 * kPi * x7), vy = vr * std::sin(2.0 * kPi * x7);
        P[3 * k] = x * len; P[3 * k + 1] = y * len; P[3 * k + 2] = z * len;
        V[3 * k] = vx * vsc; V[3 * k + 1] = vy * vsc; V[3 * k + 2] = vz * vsc;
        for (int c = 0; c < 3; ++c) { cp[c] += P[3 * k + c]; cv[c] += V[3 * k + c]; }
    }
    for (int c = 0; c < 3; ++c) { cp[c] /= double(n); cv[c] /= double(n); }
    const double m = 1.0 / double(n);
    for (size_t k = 0; k < n; ++k) {
        double p[3] = {P[3 * k] - cp[0], P[3 * k + 1] - cp[1], P[3 * k + 2] - cp[2]};
        double v[3] = {V[3 * k] - cv[0], V[3 * k + 1] - cv[1], V[3 * k + 2] - cv[2]};
        put(aos, k, stride, p, v, m, wide);
    }
    return NBODY_OK;
}

extern "C" int nbody_ic_plummer(void* aos, size_t n, size_t stride, uint64_t seed) { return ic_plummer(aos, n, stride, seed, false); }
extern "C" int nbody_ic_plummer_f64(void* aos, size_t n, size_t stride, uint64_t seed) { return ic_plummer(aos, n, stride, seed, true); }

static int ic_disc(void* aos, size_t n_disc, size_t stride, uint64_t seed, bool wide) {
    if (!aos || stride < (wide ? 80u : 40u)) return NBODY_ERR_INVALID;
    const double kPi = 3.14159265358979323846;
    Xoshiro256ss rng(seed);
    const double zero[3] = {0, 0, 0};
    put(aos, 0, stride, zero, zero, 1.0, wide);        // main.rs:52-57
    const double box_width = 10.0;                      // :59
    const double disc_mass = 2e-1;                      // :61
    const double disc_max = box_width / 2.0 / 1.2;      // :62
    const double disc_min = box_width / 10.0;           // :63
    for (size_t k = 0; k < n_disc; ++k) {               // :67-89
        double a = std::pow((std::pow(disc_max, -0.5) - std::pow(disc_min, -0.5)) * rng.uniform() + std::pow(disc_min, -0.5), -2.0);
        double phi = rng.uniform() * 2.0 * kPi;
        double x = a * std::cos(phi), y = a * std::sin(phi);
        double z = a * rng.uniform() * 0.001 - 0.0005;
        double mu = 1.0 + disc_mass * (std::pow(a, -1.5) - std::pow(disc_min, -1.5)) /
                              (std::pow(disc_max, -1.5) - std::pow(disc_min, -1.5));
        double vkep = std::sqrt(mu * 1.0 / a);
        double pos[3] = {x, y, z};
        double vel[3] = {vkep * std::sin(phi), -vkep * std::cos(phi), 0.0};
        put(aos, k + 1, stride, pos, vel, disc_mass / double(n_disc), wide);
    }
    return NBODY_OK;
}

extern "C" int nbody_ic_disc(void* aos, size_t n_disc, size_t stride, uint64_t seed) { return ic_disc(aos, n_disc, stride, seed, false); }
extern "C" int nbody_ic_disc_f64(void* aos, size_t n_disc, size_t stride, uint64_t seed) { return ic_disc(aos, n_disc, stride, seed, true); }
