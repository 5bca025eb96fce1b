// op_count.hpp — operation-counting scalar for the CPU oracle (TEST INFRASTRUCTURE ONLY).
// Running the oracle's templates on `Counted` yields the ALGORITHMIC operation count of
// one SQP-RTI step (dense restatement, no sparsity exploited) that DESIGN.md and bench.py
// use for roofline.achieved.  Thread-local counters.
#pragma once
#include <cmath>

namespace opcount {

struct Counters {
    long long add = 0, mul = 0, div = 0, sqrt = 0, trig = 0, cmp = 0;
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

inline Counted operator+(Counted a, Counted b) { get().add++; return Counted(a.v + b.v); }
inline Counted operator-(Counted a, Counted b) { get().add++; return Counted(a.v - b.v); }
inline Counted operator-(Counted a) { return Counted(-a.v); }  // sign flip: free
inline Counted operator*(Counted a, Counted b) { get().mul++; return Counted(a.v * b.v); }
inline Counted operator/(Counted a, Counted b) { get().div++; return Counted(a.v / b.v); }
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
