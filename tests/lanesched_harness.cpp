// CPU harness for the lock-step lanes' coroutine scheduler (LaneSched, csrc/fit_common.h) and the cooperative wait primitive's
// contract: tasks interleave at yields in round-robin order, a task that opens an allocation scope must not be switched out,
// every task runs to completion even when one fails, and the first error (code and message) is what run() returns.
// Built by tests/test_host_mirror_cpu.py with hipcc (host code only; no GPU call is made) against libmendeliht_hip.so.
#include "fit_common.h"
#include <cstdio>
#include <string>
#include <vector>
using namespace mih;

int main()
{
    int fails = 0;
    auto expect = [&](bool ok, const char *what) { if (!ok) { printf("FAIL: %s\n", what); ++fails; } };
    // 1. interleaving: three tasks, each appends its id, yields, appends again, ...
    {
        LaneSched s;
        std::string trace;
        std::vector<std::function<int()>> tasks;
        for (int id = 0; id < 3; ++id)
            tasks.emplace_back([&trace, id]() {
                for (int step = 0; step < 3; ++step) {
                    trace.push_back((char)('a' + id));
                    if (coop_can_yield()) current_coop()->yield();
                }
                return MIH_OK;
            });
        expect(s.run(tasks) == MIH_OK, "run returns OK");
        expect(trace == "abcabcabc", ("round-robin interleaving, got " + trace).c_str());
        expect(current_coop() == nullptr, "scheduler uninstalled after run");
    }
    // 2. disabled scheduler / single task: sequential, no yields possible
    {
        LaneSched s; s.enabled = false;
        std::string trace;
        std::vector<std::function<int()>> tasks;
        for (int id = 0; id < 2; ++id)
            tasks.emplace_back([&trace, id]() { for (int k = 0; k < 2; ++k) { trace.push_back((char)('a' + id)); if (coop_can_yield()) current_coop()->yield(); } return MIH_OK; });
        expect(s.run(tasks) == MIH_OK && trace == "aabb", "disabled scheduler runs the tasks one after the other");
    }
    // 3. allocation scopes block yields (ArenaScope is thread-local state on the coroutine's stack)
    {
        LaneSched s;
        bool could = true, after = false;
        std::vector<std::function<int()>> tasks;
        tasks.emplace_back([&]() { { ArenaScope sc(nullptr); could = coop_can_yield(); } after = coop_can_yield(); return MIH_OK; });
        tasks.emplace_back([]() { return MIH_OK; });
        expect(s.run(tasks) == MIH_OK && !could && after, "no yield inside an ArenaScope, yields again after it");
    }
    // 4. errors: the failing task's code and message come back, the others still complete
    {
        LaneSched s;
        int completed = 0;
        std::vector<std::function<int()>> tasks;
        tasks.emplace_back([&]() { current_coop()->yield(); ++completed; return MIH_OK; });
        tasks.emplace_back([&]() { set_error("boom %d", 7); return (int)MIH_NAN_LOGL; });
        tasks.emplace_back([&]() { current_coop()->yield(); current_coop()->yield(); set_error("later noise"); ++completed; return MIH_OK; });
        const int rc = s.run(tasks);
        char buf[128]; mih_last_error(buf, sizeof(buf));
        expect(rc == MIH_NAN_LOGL, "first error code returned");
        expect(std::string(buf) == "boom 7", (std::string("first error message kept, got ") + buf).c_str());
        expect(completed == 2, "the other tasks ran to completion");
    }
    // 5. deep stacks and many tasks, stacks reused over rounds
    {
        LaneSched s;
        long total = 0;
        for (int round = 0; round < 50; ++round) {
            std::vector<std::function<int()>> tasks;
            for (int id = 0; id < 18; ++id)
                tasks.emplace_back([&total, id]() {
                    volatile char pad[64 * 1024]; pad[0] = (char)id; pad[sizeof(pad) - 1] = 1;        // 64 KB of the 1 MB stack
                    for (int k = 0; k < 5; ++k) current_coop()->yield();
                    total += pad[0];
                    return MIH_OK;
                });
            if (s.run(tasks) != MIH_OK) ++fails;
        }
        expect(total == 50L * (17 * 18 / 2), "50 rounds x 18 tasks");
        expect(s.stacks.size() == 18, "stacks are reused from round to round");
    }
    printf(fails ? "lanesched: %d FAILURES\n" : "lanesched: OK\n", fails);
    return fails ? 1 : 0;
}
