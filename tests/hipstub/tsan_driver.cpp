// tsan_driver.cpp — drives the HOST state machine of liblynse_hip (built with -fsanitize=thread against tests/hipstub/hipstub.cpp)
// through the C-ABI from several threads.  TEST INFRASTRUCTURE.  Kernels do not run here: results are empty, the locks / leases /
// tickets / guards / bounded waits are what is under test.  Scenarios mirror tests/test_gpu_concurrent_readers.py, test_gpu_inflight.py,
// test_gpu_ivf_inflight.py (insert vs tickets) and the item "bound every wait that ends in a collective" (VERDICT r4 item 6).
// Exit code 0 = every scenario behaved; ThreadSanitizer reports go to stderr (the pytest wrapper fails on any).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lynse_hip.h"

static int g_fail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "CHECK FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static std::string last_error() { char b[512]; lynse_hip_last_error(b, sizeof b); return b; }

static std::vector<float> rows_of(uint64_t n, uint32_t dim, uint32_t seed) {
    std::vector<float> v((size_t)n * dim);
    uint32_t x = seed * 2654435761u + 1u;
    for (auto& f : v) { x = x * 1664525u + 1013904223u; f = (float)(x >> 8) * (1.0f / 16777216.0f); }
    return v;
}

struct Out {
    std::vector<uint64_t> rows; std::vector<float> dists; std::vector<uint32_t> counts;
    Out(uint64_t nq, uint32_t k) : rows(nq * k), dists(nq * k), counts(nq) {}
};

// ---- 1. concurrent readers on one shard: mixed batch shapes and metrics, a writer that must wait its turn
static void scenario_readers() {
    const uint32_t dim = 64; const uint64_t n = 70000;
    lynse_hip_flat* h = nullptr;
    CHECK(lynse_hip_flat_create(dim, 0, &h) == LYNSE_OK, "%s", last_error().c_str());
    const auto data = rows_of(n, dim, 1);
    CHECK(lynse_hip_flat_append_f32(h, data.data(), n) == LYNSE_OK, "%s", last_error().c_str());
    CHECK(lynse_hip_flat_finalize(h) == LYNSE_OK, "%s", last_error().c_str());
    const auto q = rows_of(256, dim, 2);
    std::atomic<int> errors{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 6; ++t)
        th.emplace_back([&, t]() {
            const uint64_t nqs[4] = {1, 5, 40, 256};
            for (int it = 0; it < 40; ++it) {
                const uint64_t nq = nqs[(it + t) % 4];
                Out o(nq, 10);
                const int metric = (it + t) % 3;   // ip / l2 / cosine
                if (lynse_hip_flat_search_f32(h, q.data(), nq, 10, metric, o.rows.data(), o.dists.data(), o.counts.data()) != LYNSE_OK) ++errors;
            }
        });
    th.emplace_back([&]() {   // a writer between the readers: append + the lazy rebuilds behind it
        const auto more = rows_of(512, dim, 3);
        for (int it = 0; it < 5; ++it) {
            if (lynse_hip_flat_append_f32(h, more.data(), 512) != LYNSE_OK) ++errors;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    });
    th.emplace_back([&]() {   // subset-filtered searches (masked scan: shared reader; gathered rows: exclusive) and the profile getters
        std::vector<uint64_t> subset;
        for (uint64_t r = 0; r < n; r += 3) subset.push_back(r);
        const uint64_t few[5] = {1, 7, 100, 4000, 69999};
        lynse_hip_flat_profile_enable(h, 1);
        for (int it = 0; it < 12; ++it) {
            Out o(40, 10);
            if (lynse_hip_flat_search_filtered_f32(h, q.data(), 40, 10, it % 3, subset.data(), subset.size(), o.rows.data(), o.dists.data(), o.counts.data()) != LYNSE_OK) ++errors;
            if (lynse_hip_flat_search_filtered_f32(h, q.data(), 3, 4, 0, few, 5, o.rows.data(), o.dists.data(), o.counts.data()) != LYNSE_OK) ++errors;
            lynse_hip_profile p;
            (void)lynse_hip_flat_profile_get(h, &p, it & 1);
        }
        lynse_hip_flat_profile_enable(h, 0);
    });
    th.emplace_back([&]() {   // getters under the reader lock
        for (int it = 0; it < 200; ++it) { (void)lynse_hip_flat_len(h); (void)lynse_hip_flat_hbm_bytes(h); int s; uint64_t r; (void)lynse_hip_flat_coarse_state(h, &s, &r); }
    });
    for (auto& x : th) x.join();
    CHECK(errors.load() == 0, "readers / writer returned errors: %s", last_error().c_str());
    CHECK(lynse_hip_flat_len(h) == n + 5 * 512, "len %llu", (unsigned long long)lynse_hip_flat_len(h));
    lynse_hip_flat_destroy(h);
}

// ---- 2. tickets: several threads submit / wait on one shard; a writer is refused while tickets are outstanding
static void scenario_tickets() {
    const uint32_t dim = 64; const uint64_t n = 70000;
    lynse_hip_flat* h = nullptr;
    CHECK(lynse_hip_flat_create(dim, 0, &h) == LYNSE_OK, "%s", last_error().c_str());
    const auto data = rows_of(n, dim, 4);
    CHECK(lynse_hip_flat_append_f32(h, data.data(), n) == LYNSE_OK, "%s", last_error().c_str());
    CHECK(lynse_hip_flat_finalize(h) == LYNSE_OK, "%s", last_error().c_str());
    CHECK(lynse_hip_flat_prepare(h, 0, 64) == LYNSE_OK, "%s", last_error().c_str());
    const auto q = rows_of(256, dim, 5);
    std::atomic<int> hard_errors{0}, refused{0}, waited{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 3; ++t)
        th.emplace_back([&, t]() {
            for (int it = 0; it < 30; ++it) {
                Out o1(64, 10), o2(8, 10);
                lynse_hip_ticket *t1 = nullptr, *t2 = nullptr;
                // an int8-shaped batch and one that needs the (lazy) f16 shadow, both in flight (the second used to fail)
                const int r1 = lynse_hip_flat_search_submit_f32_device(h, nullptr, q.data(), 64, 10, 0, o1.rows.data(), o1.dists.data(), o1.counts.data(), &t1);
                const int r2 = lynse_hip_flat_search_submit_f32_device(h, nullptr, q.data(), 8, 10, (it + t) % 3, o2.rows.data(), o2.dists.data(), o2.counts.data(), &t2);
                // (every context in flight is an INVALID_ARGUMENT by contract, not a failure of this test)
                if (r1 != LYNSE_OK && r1 != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
                if (r2 != LYNSE_OK && r2 != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
                if (t1) { if (lynse_hip_flat_search_wait(t1) != LYNSE_OK) ++hard_errors; ++waited; }
                if (t2) { if (lynse_hip_flat_search_wait(t2) != LYNSE_OK) ++hard_errors; ++waited; }
            }
        });
    th.emplace_back([&]() {
        const auto more = rows_of(16, dim, 6);
        for (int it = 0; it < 60; ++it) {
            const int rc = lynse_hip_flat_append_f32(h, more.data(), 16);
            if (rc == LYNSE_ERR_INVALID_ARGUMENT) ++refused;   // tickets outstanding: refused, by contract
            else if (rc != LYNSE_OK) ++hard_errors;
            std::this_thread::sleep_for(std::chrono::microseconds(300));
        }
    });
    th.emplace_back([&]() {   // blocking searches next to the tickets
        for (int it = 0; it < 40; ++it) {
            Out o(40, 10);
            const int rc = lynse_hip_flat_search_f32(h, q.data(), 40, 10, 0, o.rows.data(), o.dists.data(), o.counts.data());
            if (rc != LYNSE_OK && rc != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
        }
    });
    for (auto& x : th) x.join();
    CHECK(hard_errors.load() == 0, "ticket scenario: %d hard errors, last: %s", hard_errors.load(), last_error().c_str());
    CHECK(waited.load() > 0, "no ticket was ever in flight");
    fprintf(stderr, "[tsan_driver] tickets waited %d, appends refused while tickets were outstanding %d\n", waited.load(), refused.load());
    lynse_hip_flat_destroy(h);
}

static lynse_hip_ivf* make_ivf(uint64_t n, uint32_t dim, uint32_t nlist, uint32_t seed) {
    const auto data = rows_of(n, dim, seed);
    const auto cen = rows_of(nlist, dim, seed + 100);
    std::vector<uint32_t> asg(n);
    for (uint64_t i = 0; i < n; ++i) asg[i] = (uint32_t)(i % nlist);
    lynse_hip_ivf* h = nullptr;
    CHECK(lynse_hip_ivf_load(data.data(), n, dim, cen.data(), nlist, asg.data(), 0, 0, &h) == LYNSE_OK, "%s", last_error().c_str());
    return h;
}

// ---- 3. IVF: blocking searches, tickets, insert / delete and the getters, all on one index (the index guard)
static void scenario_ivf() {
    const uint32_t dim = 32, nlist = 64; const uint64_t n = 20000;
    lynse_hip_ivf* h = make_ivf(n, dim, nlist, 7);
    if (!h) return;
    const auto q = rows_of(64, dim, 8);
    std::atomic<int> hard_errors{0}, refused{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 2; ++t)
        th.emplace_back([&]() {
            for (int it = 0; it < 30; ++it) {
                Out o(16, 5);
                if (lynse_hip_ivf_search_f32(h, q.data(), 16, 5, 8, o.rows.data(), o.dists.data(), o.counts.data()) != LYNSE_OK) ++hard_errors;
            }
        });
    th.emplace_back([&]() {
        for (int it = 0; it < 30; ++it) {
            Out o(40, 5);
            lynse_hip_ivf_ticket* t = nullptr;
            const int rc = lynse_hip_ivf_search_submit_f32_device(h, nullptr, q.data(), 40, 5, 8, o.rows.data(), o.dists.data(), o.counts.data(), &t);
            if (rc != LYNSE_OK && rc != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
            if (t && lynse_hip_ivf_search_wait(t) != LYNSE_OK) ++hard_errors;
        }
    });
    th.emplace_back([&]() {   // IVFIndex::insert / delete replace the slab store under the guard (refused while tickets are outstanding)
        const auto more = rows_of(8, dim, 9);
        const uint64_t del[2] = {3, 5};
        for (int it = 0; it < 10; ++it) {
            int rc = lynse_hip_ivf_insert_f32(h, more.data(), 8);
            if (rc == LYNSE_ERR_INVALID_ARGUMENT) ++refused; else if (rc != LYNSE_OK) ++hard_errors;
            rc = lynse_hip_ivf_delete_rows(h, del, 2);
            if (rc == LYNSE_ERR_INVALID_ARGUMENT) ++refused; else if (rc != LYNSE_OK) ++hard_errors;
        }
    });
    th.emplace_back([&]() {   // entry points that used to read h->store without the guard
        std::vector<uint32_t> asg(8);
        const auto rows = rows_of(8, dim, 10);
        for (int it = 0; it < 40; ++it) {
            (void)lynse_hip_ivf_len(h);
            (void)lynse_hip_ivf_set_row_map(h, 1, 0);
            lynse_hip_profile p;
            (void)lynse_hip_ivf_profile_get(h, &p, 0);
            if (lynse_hip_ivf_assign_f32(h, rows.data(), 8, asg.data()) != LYNSE_OK) ++hard_errors;
            uint64_t st[3];
            (void)lynse_hip_ivf_ticket_stats(h, st);
        }
    });
    for (auto& x : th) x.join();
    CHECK(hard_errors.load() == 0, "ivf scenario: %d hard errors, last: %s", hard_errors.load(), last_error().c_str());
    fprintf(stderr, "[tsan_driver] ivf insert / delete refused while tickets were outstanding: %d\n", refused.load());
    lynse_hip_ivf_destroy(h);
}

// ---- 3b. a communicator of ONE rank: tickets from two threads share its exchange stream and result blocks; blocking sharded calls beside
static void scenario_comm1() {
    const uint32_t dim = 64; const uint64_t n = 70000;
    uint8_t id[128];
    CHECK(lynse_hip_comm_unique_id(id) == LYNSE_OK, "%s", last_error().c_str());
    lynse_hip_comm* c = nullptr;
    CHECK(lynse_hip_comm_create(id, 0, 1, 0, &c) == LYNSE_OK, "%s", last_error().c_str());
    lynse_hip_flat* h = nullptr;
    CHECK(lynse_hip_flat_create(dim, 0, &h) == LYNSE_OK, "%s", last_error().c_str());
    const auto data = rows_of(n, dim, 21);
    CHECK(lynse_hip_flat_append_f32(h, data.data(), n) == LYNSE_OK, "%s", last_error().c_str());
    CHECK(lynse_hip_flat_finalize(h) == LYNSE_OK, "%s", last_error().c_str());
    int seen = 0;
    CHECK(lynse_hip_comm_ranks_seen(c, &seen) == LYNSE_OK, "%s", last_error().c_str());
    const auto q = rows_of(64, dim, 22);
    std::atomic<int> hard_errors{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 2; ++t)
        th.emplace_back([&]() {
            for (int it = 0; it < 25; ++it) {
                Out o(64, 10);
                lynse_hip_ticket* tk = nullptr;
                const int rc = lynse_hip_flat_search_submit_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data(), &tk);
                if (rc != LYNSE_OK && rc != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
                if (tk && lynse_hip_flat_search_wait(tk) != LYNSE_OK) ++hard_errors;
            }
        });
    th.emplace_back([&]() {
        for (int it = 0; it < 10; ++it) {
            Out o(40, 10);
            const int rc = lynse_hip_flat_search_sharded_f32_device(h, c, q.data(), 40, 10, 1, o.rows.data(), o.dists.data(), o.counts.data());
            if (rc != LYNSE_OK && rc != LYNSE_ERR_INVALID_ARGUMENT) ++hard_errors;
        }
    });
    for (auto& x : th) x.join();
    CHECK(hard_errors.load() == 0, "1-rank communicator scenario: %d hard errors, last: %s", hard_errors.load(), last_error().c_str());
    {   // a LOCAL failure of a sharded submit rides the exchange: submit hands out a ticket, wait reports the error, the context comes back
        setenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT", "1", 1);
        Out o(64, 10);
        lynse_hip_ticket* tk = nullptr;
        const int rc = lynse_hip_flat_search_submit_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data(), &tk);
        CHECK(rc == LYNSE_OK && tk != nullptr, "a failing local part must still submit: rc %d (%s)", rc, last_error().c_str());
        const int wrc = tk ? lynse_hip_flat_search_wait(tk) : -1;
        CHECK(wrc == LYNSE_ERR_OUT_OF_MEMORY, "wait must report the local failure: rc %d (%s)", wrc, last_error().c_str());
        unsetenv("LYNSE_HIP_DEBUG_FAIL_SUBMIT");
        for (int it = 0; it < 10; ++it) {
            lynse_hip_ticket* t2 = nullptr;
            CHECK(lynse_hip_flat_search_submit_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data(), &t2) == LYNSE_OK, "%s", last_error().c_str());
            CHECK(t2 && lynse_hip_flat_search_wait(t2) == LYNSE_OK, "%s", last_error().c_str());
        }
        const auto more = rows_of(4, dim, 23);
        CHECK(lynse_hip_flat_append_f32(h, more.data(), 4) == LYNSE_OK, "the writer guard must be open again: %s", last_error().c_str());
    }
    lynse_hip_flat_destroy(h);
    lynse_hip_comm_destroy(c);
}

// ---- 4. a peer that is gone: every wait that ends in a collective returns LYNSE_ERR_TIMEOUT within the bound (stub RCCL, world = 2:
// its collectives never complete)
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
static void scenario_dead_peer() {
    const uint32_t dim = 64; const uint64_t n = 70000;
    uint8_t id[128];
    CHECK(lynse_hip_comm_unique_id(id) == LYNSE_OK, "%s", last_error().c_str());
    CHECK(lynse_hip_set_wait_timeout_ms(300) == LYNSE_OK, "set timeout");
    const auto q = rows_of(64, dim, 11);
    auto new_comm = [&]() { lynse_hip_comm* c = nullptr; CHECK(lynse_hip_comm_create(id, 0, 2, 0, &c) == LYNSE_OK, "%s", last_error().c_str()); return c; };
    auto new_flat = [&]() {
        lynse_hip_flat* h = nullptr;
        CHECK(lynse_hip_flat_create(dim, 0, &h) == LYNSE_OK, "%s", last_error().c_str());
        const auto data = rows_of(n, dim, 12);
        CHECK(lynse_hip_flat_append_f32(h, data.data(), n) == LYNSE_OK, "%s", last_error().c_str());
        CHECK(lynse_hip_flat_finalize(h) == LYNSE_OK, "%s", last_error().c_str());
        return h;
    };
    {   // (a) FLAT ticket: rank 1 dies between submit and wait
        lynse_hip_comm* c = new_comm(); lynse_hip_flat* h = new_flat();
        Out o(64, 10);
        lynse_hip_ticket* t = nullptr;
        CHECK(lynse_hip_flat_search_submit_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data(), &t) == LYNSE_OK, "%s", last_error().c_str());
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = t ? lynse_hip_flat_search_wait(t) : -1;
        CHECK(rc == LYNSE_ERR_TIMEOUT, "flat ticket: rc %d (%s)", rc, last_error().c_str());
        CHECK(ms_since(t0) < 3000.0, "flat ticket waited %.0f ms", ms_since(t0));
        // the handle is poisoned: writers are refused for good (the device may still be reading the store)
        const auto more = rows_of(4, dim, 13);
        CHECK(lynse_hip_flat_append_f32(h, more.data(), 4) == LYNSE_ERR_INVALID_ARGUMENT, "append after a timed-out ticket must be refused");
        lynse_hip_flat_destroy(h); lynse_hip_comm_destroy(c);
    }
    {   // (b) blocking sharded FLAT search
        lynse_hip_comm* c = new_comm(); lynse_hip_flat* h = new_flat();
        Out o(64, 10);
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = lynse_hip_flat_search_sharded_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data());
        CHECK(rc == LYNSE_ERR_TIMEOUT, "blocking sharded flat: rc %d (%s)", rc, last_error().c_str());
        CHECK(ms_since(t0) < 3000.0, "blocking sharded flat waited %.0f ms", ms_since(t0));
        // the communicator is marked failed: the next sharded call returns at once (it would queue behind the hung collective and wait
        // out the bound again), blocking entry point and submit alike
        const auto t1 = std::chrono::steady_clock::now();
        const int rc2 = lynse_hip_flat_search_sharded_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data());
        CHECK(rc2 == LYNSE_ERR_DEVICE && ms_since(t1) < 100.0, "second call on a timed-out communicator: rc %d after %.0f ms (%s)", rc2, ms_since(t1), last_error().c_str());
        lynse_hip_ticket* t2 = nullptr;
        const int rc3 = lynse_hip_flat_search_submit_f32_device(h, c, q.data(), 64, 10, 0, o.rows.data(), o.dists.data(), o.counts.data(), &t2);
        CHECK(rc3 == LYNSE_ERR_DEVICE && t2 == nullptr, "submit on a timed-out communicator: rc %d", rc3);
        lynse_hip_flat_destroy(h); lynse_hip_comm_destroy(c);
    }
    {   // (c) IVF ticket and (d) blocking sharded IVF search
        lynse_hip_comm* c = new_comm();
        lynse_hip_ivf* h = make_ivf(20000, 32, 64, 14);
        const auto q32 = rows_of(40, 32, 15);
        if (h) {
            Out o(40, 5);
            lynse_hip_ivf_ticket* t = nullptr;
            CHECK(lynse_hip_ivf_search_submit_f32_device(h, c, q32.data(), 40, 5, 8, o.rows.data(), o.dists.data(), o.counts.data(), &t) == LYNSE_OK, "%s", last_error().c_str());
            auto t0 = std::chrono::steady_clock::now();
            int rc = t ? lynse_hip_ivf_search_wait(t) : -1;
            CHECK(rc == LYNSE_ERR_TIMEOUT, "ivf ticket: rc %d (%s)", rc, last_error().c_str());
            CHECK(ms_since(t0) < 3000.0, "ivf ticket waited %.0f ms", ms_since(t0));
            lynse_hip_ivf_destroy(h);
            lynse_hip_comm_destroy(c);
            c = new_comm();
            h = make_ivf(20000, 32, 64, 16);
            t0 = std::chrono::steady_clock::now();
            rc = lynse_hip_ivf_search_sharded_f32_device(h, c, q32.data(), 40, 5, 8, o.rows.data(), o.dists.data(), o.counts.data());
            CHECK(rc == LYNSE_ERR_TIMEOUT, "blocking sharded ivf: rc %d (%s)", rc, last_error().c_str());
            CHECK(ms_since(t0) < 3000.0, "blocking sharded ivf waited %.0f ms", ms_since(t0));
            lynse_hip_ivf_destroy(h);
        }
        lynse_hip_comm_destroy(c);
    }
    {   // (e) the all-reduce of the sharded k-means (ShardedIvf.train): the first reduction never comes back
        lynse_hip_comm* c = new_comm();
        const uint64_t n_global = 4000, n_local = 2000;
        const auto rows = rows_of(n_local, 16, 17);
        std::vector<float> cen(8 * 16);
        std::vector<uint32_t> asg(n_local);
        uint32_t k = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = lynse_hip_ivf_kmeans_sharded(rows.data(), n_local, 0, n_global, 0, 2, 16, 8, 3, 1, 0, c, nullptr, nullptr, cen.data(), asg.data(), &k);
        CHECK(rc == LYNSE_ERR_TIMEOUT, "sharded k-means: rc %d (%s)", rc, last_error().c_str());
        CHECK(ms_since(t0) < 3000.0, "sharded k-means waited %.0f ms", ms_since(t0));
        lynse_hip_comm_destroy(c);
    }
    {   // (f) the communicator's self-check
        lynse_hip_comm* c = new_comm();
        int seen = -1;
        const int rc = lynse_hip_comm_ranks_seen(c, &seen);
        CHECK(rc == LYNSE_ERR_TIMEOUT, "ranks_seen: rc %d (%s)", rc, last_error().c_str());
        lynse_hip_comm_destroy(c);
    }
    lynse_hip_set_wait_timeout_ms(0);
}

int main(int argc, char** argv) {
    const std::string which = argc > 1 ? argv[1] : "all";
    if (which == "all" || which == "readers") scenario_readers();
    if (which == "all" || which == "tickets") scenario_tickets();
    if (which == "all" || which == "ivf") scenario_ivf();
    if (which == "all" || which == "comm1") scenario_comm1();
    if (which == "all" || which == "dead_peer") scenario_dead_peer();
    fprintf(stderr, "[tsan_driver] %s: %d failed checks\n", which.c_str(), g_fail);
    return g_fail ? 1 : 0;
}
