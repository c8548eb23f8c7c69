// staging.hpp -- pinned staging buffers of the lock-step device engine (batchsolve_dev.cpp) with their ordering made STRUCTURAL.
//
// The engine keeps ~16 pinned host buffers that the host fills and an asynchronous copy (or a kernel, through the mapped address)
// reads later -- or the reverse -- on one of two streams, every round.  Round 4 had a real ordering bug of exactly that kind (commit
// e52cc58: the advance loop refilled a shared pinned list before the asynchronous copy of its previous contents had run; found by a
// test, fixed by review).  Here the rule "the host touches a staging buffer only after the last asynchronous operation on it has
// completed" is enforced by the types instead:
//
//   * StreamClock  what the host KNOWS about a stream's progress: every asynchronous operation enqueued on it takes a ticket, a
//                  synchronise retires every ticket issued so far.  No HIP call is needed to know that an operation is complete
//                  when its stream has been synchronised since -- which is the case by construction at every use in the engine.
//   * Staged       a pinned buffer.  The ONLY ways to its memory are host() -- the host is about to read or write it -- and
//                  async_source() / async_target() -- an asynchronous operation on a stream is about to be enqueued with it, which
//                  stamps the buffer with that stream's next ticket.  host() on a buffer whose ticket has not been retired is the
//                  bug class above: by default it synchronises the stream first (correct, and visible in a trace as a sync that
//                  should not be there); in STRICT mode (SOCP_STAGING_STRICT=1, the test suite's setting; -DSOCP_STAGING_STRICT
//                  builds) it reports the buffer and aborts, so that a violation cannot hide behind the fallback.
//
// The backend (how a stream is synchronised) is a template parameter: the engine uses HipBackend, tests/cpp/staging_discipline.cpp
// a recording fake -- the rule is checked on the CPU, no GPU needed, including the e52cc58 sequence.
#pragma once
#include <cstddef>
#include <cstdio>
#include <cstdlib>

namespace socp {
namespace staging {

inline bool strict_mode()
{
#ifdef SOCP_STAGING_STRICT
    return true;
#else
    static const bool on = [] { const char *e = std::getenv("SOCP_STAGING_STRICT"); return e && e[0] == '1'; }();
    return on;
#endif
}

template <class Backend>
class StreamClock {
public:
    using stream_type = typename Backend::stream_type;
    StreamClock() = default;
    explicit StreamClock(stream_type s) : st_(s) {}
    void bind(stream_type s) { st_ = s; }
    stream_type stream() const { return st_; }
    unsigned long long ticket() { return ++issued_; }
    bool retired(unsigned long long t) const { return t <= retired_; }
    // every operation enqueued so far is complete when this returns true
    bool synchronize()
    {
        const unsigned long long upto = issued_;
        const bool ok = Backend::synchronize(st_);
        if (ok) retired_ = upto;
        return ok;
    }
    unsigned long long forced_syncs = 0;       // host() calls that had to synchronise (non-strict mode): 0 in a correct engine

private:
    stream_type st_{};
    unsigned long long issued_ = 0, retired_ = 0;
};

template <class Backend>
class Staged {
public:
    using Clock = StreamClock<Backend>;
    explicit Staged(const char *name = "?") : name_(name) {}
    void set_memory(void *p) { p_ = p; clock_ = nullptr; ticket_ = 0; }
    // The host is about to read or write the buffer.
    void *host()
    {
        if (clock_ && !clock_->retired(ticket_)) {
            if (strict_mode()) {
                std::fprintf(stderr, "[socp staging] host access to pinned buffer '%s' while an asynchronous operation on it may still be running "
                                     "(ticket %llu of its stream not retired): the engine must synchronise that stream first\n", name_, ticket_);
                std::abort();
            }
            clock_->forced_syncs++;
            (void)clock_->synchronize();
        }
        clock_ = nullptr;
        return p_;
    }
    template <class T> T *host_as() { return static_cast<T *>(host()); }
    // An asynchronous operation that READS the buffer (a host-to-device copy, a kernel reading the mapped address) is about to be
    // enqueued on c's stream / one that WRITES it (a device-to-host copy).  The same bookkeeping; two names so that a call site says what
    // it does.  A second operation on another stream while the first is pending would need both streams' tickets: the engine never
    // does that, and it is refused here rather than half-tracked.
    void *async_source(Clock &c) { return stamp(c); }
    void *async_target(Clock &c) { return stamp(c); }
    bool pending() const { return clock_ && !clock_->retired(ticket_); }
    const char *name() const { return name_; }

private:
    void *stamp(Clock &c)
    {
        if (clock_ && clock_ != &c && !clock_->retired(ticket_)) {
            std::fprintf(stderr, "[socp staging] pinned buffer '%s' handed to a second stream while an operation on the first is pending\n", name_);
            std::abort();
        }
        clock_ = &c;
        ticket_ = c.ticket();
        return p_;
    }
    const char *name_;
    void *p_ = nullptr;
    Clock *clock_ = nullptr;
    unsigned long long ticket_ = 0;
};

}  // namespace staging
}  // namespace socp
