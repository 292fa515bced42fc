// tests/cpp/canonical_pacing_test.cpp -- the HOST logic of `canonical` running ahead of its pressure callback
// (include/wayverb_amd/waveguide.h: run_device_observed, detail::field_guard), on the CPU.
//
// Test infrastructure: this translation unit supplies its OWN definitions of the C ABI entry points the header calls -- a stand-in
// engine whose "field" is a closed-form function of (step, node), so that what a callback sees can be checked against what it must
// see without a GPU -- and is NOT linked against libwayverb_amd.so.  What is under test is the header: batches of 1, 2, 4 ... 256
// steps, checkpoints before them, a look at the field of a passed step -> rollback + re-run up to it, one step per batch while
// somebody keeps looking, the callbacks' order and arguments (canonical.h:66-69), the flag protocol (waveguide.h:102-118) after the
// callbacks of the good steps, keep_going, an engine without room for checkpoints, and last_run_stats().
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>

#include <time.h>

#include "wayverb_amd/waveguide.h"

// ---- the stand-in engine ---------------------------------------------------------------------------------------------
struct wv_engine {
    uint64_t steps_done = 0, checkpoint = ~0ull;
    std::vector<uint64_t> receivers;
    std::vector<double> signal;
    uint64_t nodes = 0;
    std::vector<double> log;        // receiver rows of every completed step since the receivers were set
    uint64_t recv_first_step = 0;
    size_t log_at_checkpoint = 0;
    uint64_t flag_at_step = ~0ull;  // wv_run stops at this step with WV_FLAG_NAN
    bool no_checkpoints = false;
    // counters the tests look at
    uint64_t runs = 0, steps_run = 0, checkpoints = 0, rollbacks = 0, field_reads = 0, longest_run = 0, receiver_sets = 0, value_reads = 0;
};
static wv_engine* g_last_engine = nullptr;
static uint64_t g_flag_at_step = ~0ull;
static bool g_no_checkpoints = false;
static double g_seconds_per_step = 0;  // wv_run takes this long per step (a big mesh): the header bounds its batches by time

// the pre-update `current` of step s at node i (what waveguide.h:121 hands to `post`): any function that tells steps and nodes apart
static float field_value(uint64_t s, uint64_t i) { return (float)((s * 131u + i * 7u) % 8191u) * 0.125f + (float)s; }

extern "C" {
const char* wv_last_error(void) { return "stand-in engine"; }
void wv_default_options(wv_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(*o);
}
int wv_create(const wv_mesh* m, const wv_options*, wv_engine** out) {
    auto* e = new wv_engine;
    e->nodes = (uint64_t)m->nx * m->ny * m->nz;
    e->flag_at_step = g_flag_at_step;
    e->no_checkpoints = g_no_checkpoints;
    g_last_engine = e;
    *out = e;
    return WV_OK;
}
void wv_destroy(wv_engine* e) {
    if (g_last_engine == e) g_last_engine = nullptr;
    delete e;
}
int wv_set_source(wv_engine* e, int, uint64_t, const double* s, uint64_t n) {
    e->signal.assign(s, s + n);
    return WV_OK;
}
int wv_set_receivers(wv_engine* e, const uint64_t* nodes, uint32_t n) {
    e->receivers.assign(nodes, nodes + n);
    e->recv_first_step = e->steps_done;  // (as the engine: rows are recorded from here on)
    e->log.clear();
    e->checkpoint = ~0ull;               // (... and a checkpoint taken with other receivers is of no use)
    ++e->receiver_sets;
    return WV_OK;
}
int wv_run(wv_engine* e, uint64_t n, uint64_t* done, int32_t* flag) {
    ++e->runs;
    uint64_t k = 0;
    *flag = 0;
    for (; k < n && e->steps_done < e->signal.size(); ++k) {
        if (e->steps_done == e->flag_at_step) {
            *flag = WV_FLAG_NAN;
            break;
        }
        for (uint64_t r : e->receivers) e->log.push_back((double)field_value(e->steps_done, r));
        ++e->steps_done;
        ++e->steps_run;
    }
    if (e->longest_run < k) e->longest_run = k;
    if (g_seconds_per_step > 0 && k) {
        const double s = g_seconds_per_step * (double)k;
        timespec ts{(time_t)s, (long)((s - (double)(time_t)s) * 1e9)};
        nanosleep(&ts, nullptr);
    }
    *done = k;
    return WV_OK;
}
int wv_fetch_receivers(wv_engine* e, uint64_t first, uint64_t n, double* dst) {
    const size_t w = e->receivers.size();
    if (first < e->recv_first_step || (first - e->recv_first_step + n) * w > e->log.size()) return WV_E_INVALID_ARGUMENT;
    std::memcpy(dst, e->log.data() + (first - e->recv_first_step) * w, (size_t)n * w * sizeof(double));
    return WV_OK;
}
int wv_checkpoint(wv_engine* e) {
    if (e->no_checkpoints) return WV_E_HIP;
    e->checkpoint = e->steps_done;
    e->log_at_checkpoint = e->log.size();
    ++e->checkpoints;
    return WV_OK;
}
int wv_rollback(wv_engine* e) {
    if (e->checkpoint == ~0ull) return WV_E_STATE;
    e->steps_done = e->checkpoint;
    e->log.resize(e->log_at_checkpoint);
    ++e->rollbacks;
    return WV_OK;
}
int wv_query(wv_engine*, int, uint64_t* v) {
    *v = 0;
    return WV_OK;
}
// WV_BUF_PREVIOUS after wv_run = the pre-update `current` of the last completed step
int wv_read_field(wv_engine* e, int which, void* dst, int elem) {
    if (which != WV_BUF_PREVIOUS || elem != 4 || e->steps_done == 0) return WV_E_INVALID_ARGUMENT;
    ++e->field_reads;
    for (uint64_t i = 0; i < e->nodes; ++i) static_cast<float*>(dst)[i] = field_value(e->steps_done - 1, i);
    return WV_OK;
}
int wv_read_value(wv_engine* e, int which, uint64_t index, double* v) {
    if (which != WV_BUF_PREVIOUS || e->steps_done == 0) return WV_E_INVALID_ARGUMENT;
    ++e->value_reads;
    *v = (double)field_value(e->steps_done - 1, index);
    return WV_OK;
}
// (referenced by parts of the header this test does not exercise)
int wv_write_value(wv_engine*, int, uint64_t, double) { return WV_E_STATE; }
int wv_write_field(wv_engine*, int, const void*, int) { return WV_E_STATE; }
int wv_step(wv_engine*, int32_t*) { return WV_E_STATE; }
int wv_swap(wv_engine*) { return WV_E_STATE; }
int wv_make_box_nodes(int32_t nx, int32_t ny, int32_t nz, int32_t, int32_t, int32_t, int32_t, wv_condensed_node* nodes, uint64_t counts[3]) {
    for (int64_t i = 0; i < (int64_t)nx * ny * nz; ++i) nodes[i] = wv_condensed_node{WV_ID_INSIDE, 0};  // (every node a legal source / receiver place)
    counts[0] = counts[1] = counts[2] = 0;
    return WV_OK;
}
}  // extern "C"

#define REQUIRE(cond)                                                       \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("REQUIRE failed: %s (line %d)\n", #cond, __LINE__); \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

using namespace wayverb;

namespace {
struct context {
    int device = -1;
};
const waveguide::vec3 kSource{0.1f, 0.1f, 0.1f}, kReceiver{0.2f, 0.15f, 0.1f};

// a run of `steps` steps whose callback looks at the field whenever `looks(step)` says so; returns what it saw
struct seen {
    std::vector<size_t> steps, order;
    bool all_right = true;
};
template <typename Looks>
seen run(size_t steps, Looks looks, waveguide::run_stats* stats, wv_engine* counters, const std::atomic_bool* keep = nullptr, bool expect_done = true) {
    const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
    const core::environment env{};
    const double sr = waveguide::compute_sample_rate(mesh.get_descriptor(), env.speed_of_sound);
    const std::atomic_bool go{true};
    seen s;
    const size_t nodes = waveguide::compute_num_nodes(mesh.get_descriptor());
    wv_engine snapshot;
    const auto out = waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6},
                                          ((double)steps - 0.5) / sr, keep ? *keep : go, [&](auto& queue, const auto& buffer, auto step, auto total) {
                                              s.all_right = s.all_right && total == steps;
                                              s.order.push_back(step);
                                              if (looks(step)) {
                                                  const auto field = core::read_from_buffer<float>(queue, buffer);
                                                  s.all_right = s.all_right && field.size() == nodes;
                                                  for (size_t i = 0; i < field.size(); i += 37) s.all_right = s.all_right && field[i] == field_value(step, i);
                                                  s.all_right = s.all_right && core::read_value<float>(queue, buffer, 5) == field_value(step, 5);
                                                  s.steps.push_back(step);
                                              }
                                              if (g_last_engine) snapshot = *g_last_engine;
                                          });
    REQUIRE(bool(out) == expect_done);
    if (out) {
        // the records are the receiver's samples of every step, in order (directional_receiver.cpp:29-67 on the stand-in's values)
        const auto& d = out->front().band.directional;
        REQUIRE(d.size() == steps);
        const size_t r = waveguide::compute_index(mesh.get_descriptor(), kReceiver);
        for (size_t k = 0; k < steps; ++k) REQUIRE(d[k].pressure == field_value(k, r));
    }
    if (stats) *stats = waveguide::last_run_stats();
    if (counters) *counters = snapshot;
    return s;
}
}  // namespace

int main() {
    waveguide::run_stats st;
    wv_engine eng;
    // ---- nobody looks: batches of 1, 2, 4 ... 256, a checkpoint before every batch of more than one step, no rollback, every step run once
    {
        const auto s = run(1000, [](size_t) { return false; }, &st, &eng);
        REQUIRE(s.all_right && s.order.size() == 1000 && s.steps.empty());
        for (size_t k = 0; k < 1000; ++k) REQUIRE(s.order[k] == k);
        REQUIRE(st.steps == 1000 && st.rollbacks == 0 && st.steps_rerun == 0 && st.fields_looked_at == 0);
        REQUIRE(st.batches == 11 && st.checkpoints == 10);  // 1 + 2 + ... + 256 = 511, then 256, then 233
        REQUIRE(eng.steps_run <= 1000 && eng.field_reads == 0);
    }
    // ---- somebody looks at every step: one step per batch from the start, nothing is ever run twice
    {
        const auto s = run(300, [](size_t) { return true; }, &st, &eng);
        REQUIRE(s.all_right && s.steps.size() == 300);
        REQUIRE(st.batches == 300 && st.rollbacks == 0 && st.checkpoints == 0 && st.fields_looked_at == 300);
    }
    // ---- a look every 7 steps (never 16 steps without one): the run never gets ahead of a look; once the interval is known the
    // batches end on the steps looked at
    {
        const auto s = run(300, [](size_t k) { return k % 7 == 0; }, &st, &eng);
        REQUIRE(s.all_right && s.steps.size() == 43 && st.rollbacks == 0 && st.steps_rerun == 0 && st.batches < 80);
    }
    // ---- a look every 40 / 97 / 511 steps: the run gets ahead, comes back for exactly those steps, and the callbacks still fire once
    // each, in order
    for (size_t every : {size_t{40}, size_t{97}, size_t{511}}) {
        const auto s = run(1200, [every](size_t k) { return k % every == every - 1; }, &st, &eng);
        REQUIRE(s.all_right && s.order.size() == 1200 && s.steps.size() == 1200 / every);
        for (size_t k = 0; k < 1200; ++k) REQUIRE(s.order[k] == k);
        for (size_t j = 0; j < s.steps.size(); ++j) REQUIRE(s.steps[j] == (j + 1) * every - 1);
        // two looks tell the interval (a rollback each, when the run was ahead); from then on every look falls on the last step of a batch
        REQUIRE(st.rollbacks >= 1 && st.rollbacks <= 2 && st.fields_looked_at == s.steps.size());
        REQUIRE(st.steps_rerun <= 2 * every && st.batches <= 40 + 1200 / every + 1200 / 256);
        std::printf("a look every %zu steps of 1200: %zu batches, %zu checkpoints, %zu rollbacks, %zu steps re-run\n", every, st.batches,
                    st.checkpoints, st.rollbacks, st.steps_rerun);
    }
    // ---- an observer that changes its mind: regular for a while, then irregular, then regular at another interval
    {
        const auto irregular = [](size_t k) { return (k < 400 && k % 25 == 24) || k == 431 || k == 507 || k == 520 || (k >= 600 && k % 60 == 0); };
        const auto s = run(1200, irregular, &st, &eng);
        size_t expected = 0;
        for (size_t k = 0; k < 1200; ++k) expected += irregular(k);
        REQUIRE(s.all_right && s.steps.size() == expected && s.order.size() == 1200 && st.fields_looked_at == expected);
        std::printf("an observer that changes its mind: %zu looks, %zu batches, %zu rollbacks, %zu steps re-run\n", expected, st.batches, st.rollbacks,
                    st.steps_rerun);
    }
    // ---- the last step of a batch is on the device as it is: no rollback for it (steps 0, 2, 6, 14, 30 ... end batches of 1, 2, 4 ...)
    {
        const auto s = run(600, [](size_t k) { return k == 14; }, &st, &eng);
        REQUIRE(s.all_right && s.steps.size() == 1 && st.rollbacks == 0);
    }
    // ---- an engine without room for checkpoints: one step per batch, like the reference's loop; looks cost nothing extra
    {
        g_no_checkpoints = true;
        const auto s = run(200, [](size_t k) { return k % 50 == 3; }, &st, &eng);
        g_no_checkpoints = false;
        REQUIRE(s.all_right && s.steps.size() == 4 && st.batches == 200 && st.checkpoints == 0 && st.rollbacks == 0);
    }
    // ---- batches are bounded by time, not only by count: an engine that takes 2 ms per step (a 800^3 mesh) never gets more than
    // ~25 steps (50 ms, waveguide::batch_seconds()) at a time, so that keep_going and the progress callbacks stay that close to the
    // reference's per-step cadence; a cancel is honoured within the batch in flight
    {
        g_seconds_per_step = 0.002;
        const auto s = run(300, [](size_t) { return false; }, &st, &eng);
        REQUIRE(s.all_right && s.order.size() == 300 && eng.longest_run <= 32 && st.batches >= 300 / 32);
        std::atomic_bool keep{true};
        const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
        const core::environment env{};
        const double sr = waveguide::compute_sample_rate(mesh.get_descriptor(), env.speed_of_sound);
        size_t calls = 0;
        double cancelled_at = 0, returned_at = 0;
        const auto out = waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6}, 499.5 / sr, keep,
                                              [&](auto&, const auto&, auto step, auto) {
                                                  ++calls;
                                                  if (step == 200) {
                                                      keep = false;
                                                      cancelled_at = waveguide::detail::seconds_now();
                                                  }
                                              });
        returned_at = waveguide::detail::seconds_now();
        g_seconds_per_step = 0;
        REQUIRE(!out && calls == 201);
        REQUIRE(returned_at - cancelled_at < 2 * 0.05 + 0.02);  // (the re-run up to the cancelling step is at most one more batch)
        std::printf("2 ms per step: batches of at most %llu steps; cancel honoured after %.0f ms\n", (unsigned long long)eng.longest_run,
                    1e3 * (returned_at - cancelled_at));
    }
    // ---- keep_going turned off by the callback: no later step's callback fires, the engine is back where that step left it, canonical
    // returns nothing (canonical.h:84-87; waveguide.h:80 tests keep_going before every iteration)
    {
        std::atomic_bool keep{true};
        const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
        const core::environment env{};
        const double sr = waveguide::compute_sample_rate(mesh.get_descriptor(), env.speed_of_sound);
        size_t calls = 0;
        const auto out = waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6}, 499.5 / sr, keep,
                                              [&](auto&, const auto&, auto step, auto) {
                                                  ++calls;
                                                  if (step == 100) keep = false;
                                              });
        REQUIRE(!out && calls == 101);
        REQUIRE(g_last_engine == nullptr);  // (the engine is gone with the run; what it was at the end:)
        REQUIRE(waveguide::last_run_stats().steps == 101);
    }
    // ---- a flag on step 77: the callbacks of steps 0 .. 76 fire, then the reference's exception (waveguide.h:102-118); a callback that
    // looks at a step of the very batch that met the flag still gets its field (none of that batch's fields is on the device)
    for (int look : {0, 1}) {
        g_flag_at_step = 77;
        size_t calls = 0;
        bool threw = false, right = true;
        const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
        const core::environment env{};
        const double sr = waveguide::compute_sample_rate(mesh.get_descriptor(), env.speed_of_sound);
        const std::atomic_bool go{true};
        try {
            (void)waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6}, 299.5 / sr, go,
                                       [&](auto& queue, const auto& buffer, auto step, auto) {
                                           right = right && step == calls;
                                           ++calls;
                                           if (look && step == 76) right = right && core::read_value<float>(queue, buffer, 9) == field_value(76, 9);
                                       });
        } catch (const core::exceptions::value_is_nan&) {
            threw = true;
        }
        g_flag_at_step = ~0ull;
        REQUIRE(threw && right && calls == 77);
    }
    // ---- waveguide::run<pre, post> with one of this header's sources (bin/boundary_test/boundary_test.cpp:137-147: a soft source and a
    // lambda around callback_accumulator<postprocessor::node>s): the steps are taken in batches ahead of `post`, which fires once per
    // step, in order, and reads what the per-step loop would have read -- the nodes it reads are learned in the first step and served
    // from recorded rows; a node it has never read before (step 300) still gets its step's value
    {
        const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
        const std::atomic_bool go{true};
        const size_t steps = 700;
        std::vector<float> input(steps, 0.0f);
        input[0] = 1000.0f;
        auto prep = waveguide::preprocessor::make_soft_source(33, input.begin(), input.end());
        std::vector<waveguide::postprocessor::node> holders{waveguide::postprocessor::node{5}, waveguide::postprocessor::node{101},
                                                             waveguide::postprocessor::node{230}};
        std::vector<std::vector<float>> outputs(holders.size());
        size_t counter = 0;
        bool right = true;
        wv_engine snapshot;
        const size_t done = waveguide::run(context{}, mesh, prep,
                                           [&](auto& queue, const auto& buffer, auto step) {
                                               right = right && step == counter++;
                                               for (size_t k = 0; k < holders.size(); ++k) outputs[k].push_back(holders[k](queue, buffer, step));
                                               if (step >= 300 && step % 100 == 0)
                                                   right = right && core::read_value<float>(queue, buffer, 77) == field_value(step, 77);
                                               if (g_last_engine) snapshot = *g_last_engine;
                                           },
                                           go);
        const auto stats = waveguide::last_run_stats();
        REQUIRE(done == steps && right && counter == steps && prep.begin() == prep.end());
        for (size_t k = 0; k < holders.size(); ++k) {
            REQUIRE(outputs[k].size() == steps);
            for (size_t t = 0; t < steps; ++t) REQUIRE(outputs[k][t] == field_value(t, holders[k].get_output_node()));
        }
        // step 0 teaches the three nodes, step 300 the fourth (a rollback: the run was ahead); everything else comes from the rows
        REQUIRE(stats.batches < 40 && stats.rollbacks == 1 && snapshot.receiver_sets == 3 && snapshot.value_reads == 4);
        REQUIRE(stats.reads_missed == 4 && stats.reads_served == 3 * (steps - 1) + 3);  // (node 77: missed at step 300, served at 400, 500, 600)
        std::printf("run<soft_source, post>: %zu steps in %zu batches, %zu reads served from recorded rows, %zu from the engine, %zu rollback(s)\n", done,
                    stats.batches, stats.reads_served, stats.reads_missed, stats.rollbacks);
        // cancelled from `post`: no later callback, the source object one sample further than the steps done (waveguide.h:80)
        std::atomic_bool keep{true};
        auto prep2 = waveguide::preprocessor::make_hard_source(33, input.begin(), input.end());
        size_t calls = 0;
        const size_t done2 = waveguide::run(context{}, mesh, prep2,
                                            [&](auto& queue, const auto& buffer, auto step) {
                                                ++calls;
                                                (void)core::read_value<float>(queue, buffer, 5);
                                                if (step == 150) keep = false;
                                            },
                                            keep);
        REQUIRE(done2 == 151 && calls == 151 && (size_t)std::distance(input.begin(), prep2.begin()) == 152);
        // a lambda around the source (src/waveguide/tests/waveguide_tests.cpp:95-100) cannot be recognised: the reference's loop, step by step
        // (the stand-in engine has no wv_step: reaching it is the proof)
        auto prep3 = waveguide::preprocessor::make_soft_source(33, input.begin(), input.end());
        bool stepped = false;
        try {
            (void)waveguide::run(context{}, mesh, [&](auto& queue, auto& buffer, auto step) { return prep3(queue, buffer, step); },
                                 [](auto&, const auto&, auto) {}, go);
        } catch (const waveguide::engine_error&) {
            stepped = true;
        }
        REQUIRE(stepped);
    }
    // ---- a callback that cannot see the field, or promises not to look: whole batches (after a first one of 8 steps that finds out what
    // a step costs), no checkpoints at all
    {
        const auto mesh = waveguide::make_box_mesh(8, 7, 6, 0.05f, waveguide::to_flat_coefficients(0.1));
        const core::environment env{};
        const double sr = waveguide::compute_sample_rate(mesh.get_descriptor(), env.speed_of_sound);
        const std::atomic_bool go{true};
        size_t calls = 0;
        auto out = waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6}, 999.5 / sr, go,
                                        [&](size_t step, size_t total) { calls += step < total; });
        REQUIRE(bool(out) && calls == 1000 && waveguide::last_run_stats().batches == 5 && waveguide::last_run_stats().checkpoints == 0);
        calls = 0;
        out = waveguide::canonical(context{}, mesh, kSource, kReceiver, env, waveguide::single_band_parameters{100.0, 0.6}, 999.5 / sr, go,
                                   waveguide::progress_only([&](auto&, const auto&, auto, auto) { ++calls; }));
        REQUIRE(bool(out) && calls == 1000 && waveguide::last_run_stats().batches == 5 && waveguide::last_run_stats().checkpoints == 0);
    }
    std::puts("CANONICAL PACING OK");
    return 0;
}
