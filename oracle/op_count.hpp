// op_count.hpp — operation-counting scalar for the CPU oracle (TEST INFRASTRUCTURE ONLY).
// Running the oracle's templates on `Counted` yields the ALGORITHMIC operation count of
// one SQP-RTI step (dense restatement, no sparsity exploited) that DESIGN.md and bench.py
// use for roofline.achieved.  Thread-local counters.
// Beside the dense count a STRUCTURED count is kept (s_add, s_mul, s_div): an operation is left out
// of it when its result needs no arithmetic because of an operand's value — x + 0, x - 0, x * 0,
// x * (+-1), 0 / x, x / (+-1).  In the dense restatement the structural entries are exact: the
// Hessian blocks are diagonal (off-diagonals 0, S = 0), A = I + two entries (seven of nine entries are
// exact 0 / 1, the ERK sensitivities keep them exact), Vx / Vu are selections, the eliminated stage-0
// state contributes zeros.  So the structured count is what an implementation that exploits the
// sparsity of this OCP (diagonal H, A = I + (a, b), x0 eliminated) has to execute, measured on the
// same run as the dense one.  (Data that happens to be exactly 0 — the steering reference — is left
// out too: the structured figure errs low, i.e. against the kernel.)
#pragma once
#include <cmath>

namespace opcount {

struct Counters {
    long long add = 0, mul = 0, div = 0, sqrt = 0, trig = 0, cmp = 0;
    long long s_add = 0, s_mul = 0, s_div = 0;
};
inline Counters& get() {
    static thread_local Counters c;
    return c;
}
inline void reset() { get() = Counters(); }

struct Counted {
    double v;
    Counted() : v(0.0) {}
    explicit Counted(double x) : v(x) {}
    explicit operator double() const { return v; }
};

inline bool is0(double x) { return x == 0.0; }
inline bool is1(double x) { return x == 1.0 || x == -1.0; }
inline Counted operator+(Counted a, Counted b) { get().add++; if (!is0(a.v) && !is0(b.v)) get().s_add++; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { get().add++; if (!is0(a.v) && !is0(b.v)) get().s_add++; return Counted(a.v - b.v); }
inline Counted operator-(Counted a) { return Counted(-a.v); }  // sign flip: free
inline Counted operator*(Counted a, Counted b) {
    get().mul++;
    if (!is0(a.v) && !is0(b.v) && !is1(a.v) && !is1(b.v)) get().s_mul++;
    return Counted(a.v * b.v);
}
inline Counted operator/(Counted a, Counted b) {
    get().div++;
    if (!is0(a.v) && !is1(b.v)) get().s_div++;
    return Counted(a.v / b.v);
}
inline bool operator<(Counted a, Counted b) { get().cmp++; return a.v < b.v; }
inline bool operator>(Counted a, Counted b) { get().cmp++; return a.v > b.v; }
inline bool operator<=(Counted a, Counted b) { get().cmp++; return a.v <= b.v; }
inline bool operator>=(Counted a, Counted b) { get().cmp++; return a.v >= b.v; }
inline Counted sqrt(Counted a) { get().sqrt++; return Counted(std::sqrt(a.v)); }
inline Counted sin(Counted a) { get().trig++; return Counted(std::sin(a.v)); }
inline Counted cos(Counted a) { get().trig++; return Counted(std::cos(a.v)); }
inline Counted tan(Counted a) { get().trig++; return Counted(std::tan(a.v)); }
inline Counted fabs(Counted a) { get().cmp++; return Counted(std::fabs(a.v)); }

}  // namespace opcount
