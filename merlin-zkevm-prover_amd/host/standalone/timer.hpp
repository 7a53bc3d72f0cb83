// timer.hpp (standalone stand-in) -- TimerStart / TimerStopAndLog with the reference's phase names (timer.hpp:19-30).  Here the
// phases are timed with HIP events on the library's stream (the work is asynchronous to the host) and collected in a table that
// bench_starks.py / tests read: mi::phaseTimes().
#ifndef TIMER_HPP
#define TIMER_HPP
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>
#include "mi_runtime.hpp"
namespace mi {
struct PhaseTimer {
    std::vector<std::pair<std::string, int>> slots; // phase name -> event slot
    uint64_t minFree = ~0ULL;                       // low-water mark of free HBM, sampled at every phase end
    bool enabled = std::getenv("MI_STARK_PHASE_TIMES") != nullptr; // off by default: an event pair per phase
    int slotOf(const char *name)
    {
        for (auto &s : slots) if (s.first == name) return s.second;
        slots.push_back({name, (int)slots.size()});
        return slots.back().second;
    }
};
inline PhaseTimer &phaseTimer() { static PhaseTimer t; return t; }
inline void timerStart(const char *name) { PhaseTimer &t = phaseTimer(); if (t.enabled && t.slots.size() < 60) mi_timer_start(ctx(), t.slotOf(name)); }
inline void timerStop(const char *name)
{
    PhaseTimer &t = phaseTimer();
    if (!t.enabled) return;
    for (auto &s : t.slots) if (s.first == name) mi_timer_stop(ctx(), s.second);
    uint64_t fr = 0, tot = 0;
    if (mi_dev_mem_info(ctx(), &fr, &tot) == MI_OK && fr < t.minFree) t.minFree = fr;
}
// (phase, milliseconds) in first-start order; synchronises on each stop event
inline std::vector<std::pair<std::string, float>> phaseTimes()
{
    std::vector<std::pair<std::string, float>> out;
    for (auto &s : phaseTimer().slots) { float ms = 0; if (mi_timer_elapsed_ms(ctx(), s.second, &ms) == MI_OK) out.push_back({s.first, ms}); }
    return out;
}
} // namespace mi
#define TimerStart(name) mi::timerStart(#name)
#define TimerStop(name) mi::timerStop(#name)
#define TimerLog(name)
#define TimerStopAndLog(name) mi::timerStop(#name)
#endif
