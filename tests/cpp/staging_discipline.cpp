// staging_discipline.cpp -- the pinned-buffer rule of the lock-step device engine, checked on the CPU (no GPU, no HIP):
// socp_amd/csrc/staging.hpp with a recording fake backend.  VERDICT r4 #4.
//
//   staging_discipline ok        the sequences the engine runs: every host access finds its buffer's ticket retired, no forced sync
//   staging_discipline e52cc58   round 4's bug as it was written -- the start list copied asynchronously, the same pinned buffer
//                                refilled by the advance loop before any synchronise.  Default mode: the second host access
//                                synchronises first (counted, result correct); strict mode (SOCP_STAGING_STRICT=1): abort().
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../socp_amd/csrc/staging.hpp"

struct FakeBackend {
    using stream_type = int;
    static std::vector<int> &log() { static std::vector<int> l; return l; }
    static bool synchronize(int s) { log().push_back(s); return true; }
};
using Clock = socp::staging::StreamClock<FakeBackend>;
using Buf = socp::staging::Staged<FakeBackend>;

static int fail(const char *what) { std::printf("FAIL: %s\n", what); return 1; }

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "ok";
    int mem_list[8] = {0}, mem_status[8] = {0}, mem_x[8] = {0};
    Clock main_clk(1), fs_clk(2);
    Buf hList("hList"), hStatus("hStatus"), hX("hX"), hListF("hListF");
    int mem_f[8];
    hList.set_memory(mem_list); hStatus.set_memory(mem_status); hX.set_memory(mem_x); hListF.set_memory(mem_f);

    if (mode == "ok") {
        // one inner pass of the advance loop, twice: fill the list, copy it, launch, read the status back, synchronise, read it
        for (int round = 0; round < 2; round++) {
            std::memset(hList.host(), round, sizeof(mem_list));
            (void)hList.async_source(main_clk);              // hipMemcpyAsync H2D
            (void)hStatus.async_target(main_clk);            // hipMemcpyAsync D2H
            if (!hList.pending() || !hStatus.pending()) return fail("buffers with an operation in flight must read as pending");
            if (!main_clk.synchronize()) return fail("synchronise");
            if (hList.pending() || hStatus.pending()) return fail("a synchronise retires every ticket issued before it");
            (void)hStatus.host();
            // the evaluation part: a list on the second stream, a start from hX on the main one, both streams synchronised
            std::memset(hListF.host(), 1, sizeof(mem_f));
            (void)hListF.async_source(fs_clk);
            std::memset(hX.host(), 2, sizeof(mem_x));
            (void)hX.async_source(main_clk);                 // the start kernel reads the mapped buffer
            fs_clk.synchronize(); main_clk.synchronize();
        }
        if (main_clk.forced_syncs || fs_clk.forced_syncs) return fail("the engine's own sequences must need no forced synchronise");
        // a ticket issued AFTER a synchronise is not retired by it
        (void)hList.host();
        (void)hList.async_source(main_clk);
        if (!hList.pending()) return fail("an operation enqueued after the last synchronise is pending");
        main_clk.synchronize();
        std::printf("ok: %zu synchronises, none forced\n", FakeBackend::log().size());
        return 0;
    }
    if (mode == "e52cc58") {
        // start_chains(list): the list goes into the shared pinned buffer and an asynchronous copy of it is enqueued ...
        std::memset(hList.host(), 7, sizeof(mem_list));
        (void)hList.async_source(main_clk);
        // ... and the advance loop refills the same buffer right away (round 4 wrote `std::memcpy(hList.p, adv.data(), ...)` here)
        const size_t syncs_before = FakeBackend::log().size();
        void *p = hList.host();                              // strict mode: aborts HERE, naming the buffer
        std::memset(p, 9, sizeof(mem_list));
        if (FakeBackend::log().size() != syncs_before + 1 || FakeBackend::log().back() != 1) return fail("the violating access must synchronise the buffer's stream first");
        if (main_clk.forced_syncs != 1) return fail("... and be counted");
        std::printf("ok: the violating host access synchronised stream 1 first (forced_syncs = %llu)\n", main_clk.forced_syncs);
        return 0;
    }
    return fail("unknown mode");
}
