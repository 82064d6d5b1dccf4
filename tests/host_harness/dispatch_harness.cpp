// dispatch_harness.cpp -- drives the device-free slice of the library's host side (wdbx-py_amd/csrc/host_dispatch.h: the
// shard dispatcher, ordered locking, grow bookkeeping, option lookup, the exception barrier) hard enough for ThreadSanitizer /
// AddressSanitizer / UBSan to see its hand-overs.  TEST INFRASTRUCTURE: built and run by tests/test_host_dispatch_sanitizers.py
// with plain g++ (no HIP), never linked into the product.
//
//   dispatch_harness [dispatches=10000] [workers=8]
//
// Exit code 0 and a last line "harness ok ..." = every check passed; sanitizer reports go to stderr and change the exit code
// (halt_on_error / exitcode in the test's *SAN_OPTIONS).
#include "host_dispatch.h"

#include <cstdlib>
#include <random>
#include <stdexcept>

#define CHECK(cond)                                                               \
  do {                                                                            \
    if (!(cond)) {                                                                \
      fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
      exit(2);                                                                    \
    }                                                                             \
  } while (0)

static std::atomic<int> g_bound{0};
static void bind_count(int) { g_bound.fetch_add(1); }

// ---- 1. the dispatcher: many dispatches over W workers, jobs that succeed, fail, throw and dawdle ----
static void test_dispatcher(int dispatches, int nworkers) {
  Dispatcher d;
  std::vector<int> devs;
  for (int i = 0; i < nworkers; ++i) devs.push_back(i % 3);
  d.start(devs, bind_count);
  const int S = nworkers + 1;
  std::vector<uint64_t> per_shard(S, 0);  // written by shard s's thread only, read by the dispatcher between runs
  uint64_t expect_total = 0;
  std::mt19937 rng(7);
  for (int it = 0; it < dispatches; ++it) {
    const int kind = (int)(rng() % 16);            // 0: one shard fails, 1: one throws, 2: one is slow, else all fine
    const int victim = (int)(rng() % S);
    const uint64_t add = (uint64_t)it + 1;
    const std::function<int(int)> job = [&](int s) -> int {
      per_shard[s] += add;                         // plain (non-atomic) data handed over by posted / done only
      if (kind == 2 && s == victim) std::this_thread::sleep_for(std::chrono::microseconds(300));  // past the workers' spin window
      if (kind == 0 && s == victim) return fail(WDBX_E_HIP, "shard %d failed in dispatch %d", s, it);
      if (kind == 1 && s == victim) throw std::runtime_error("boom in shard " + std::to_string(s));
      return WDBX_OK;
    };
    const int rc = d.run(job);
    expect_total += add;
    if (kind == 0) {
      CHECK(rc == WDBX_E_HIP);
      CHECK(g_err.find("failed in dispatch " + std::to_string(it)) != std::string::npos);
    } else if (kind == 1) {
      CHECK(rc == WDBX_E_STATE);
      CHECK(g_err.find("boom in shard " + std::to_string(victim)) != std::string::npos);
    } else {
      CHECK(rc == WDBX_OK);
    }
    if (it % 257 == 0)
      for (int s = 0; s < S; ++s) CHECK(per_shard[s] == expect_total);  // every shard ran every job exactly once so far
    if (it % 1000 == 999) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // let the workers fall asleep: the cv path
  }
  for (int s = 0; s < S; ++s) CHECK(per_shard[s] == expect_total);
  CHECK(d.dispatches == (uint64_t)dispatches);
  d.stop();
  CHECK(g_bound.load() >= nworkers);
  d.stop();  // idempotent
}

// ---- 2. stop while workers spin / sleep / have just finished, many times ----
static void test_start_stop_churn() {
  for (int round = 0; round < 200; ++round) {
    Dispatcher d;
    d.start({0, 1, 2}, nullptr);
    std::atomic<int> ran{0};
    const std::function<int(int)> job = [&](int) -> int {
      ran.fetch_add(1);
      return WDBX_OK;
    };
    const int n = round % 4;
    for (int i = 0; i < n; ++i) CHECK(d.run(job) == WDBX_OK);
    CHECK(ran.load() == 4 * n);
    if (round % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(350));  // asleep on the cv when stop() comes
    // (destructor stops: spinning, sleeping and never-used workers all have to leave)
  }
}

// ---- 3. ordered locks: group-style lockers (all handles, in order) against per-handle lockers ----
static void test_ordered_locks() {
  constexpr int H = 6;
  std::vector<std::mutex> mus(H);
  std::vector<long> counters(H, 0);  // counters[h] guarded by mus[h]
  std::vector<std::mutex*> all;
  for (auto& m : mus) all.push_back(&m);
  std::vector<std::thread> ts;
  for (int t = 0; t < 3; ++t)
    ts.emplace_back([&] {
      for (int i = 0; i < 3000; ++i) {
        OrderedLocks locks(all);
        for (int h = 0; h < H; ++h) ++counters[h];
      }
    });
  for (int t = 0; t < 5; ++t)
    ts.emplace_back([&, t] {
      for (int i = 0; i < 8000; ++i) {
        const int h = (i + t) % H;
        std::lock_guard<std::mutex> lk(mus[h]);
        ++counters[h];
      }
    });
  for (auto& th : ts) th.join();
  long total = 0;
  for (long c : counters) total += c;
  CHECK(total == 3L * 3000 * H + 5L * 8000);
}

// ---- 4. grow bookkeeping: sizes, failures, nothing dangling, nothing leaked (ASan watches the heap) ----
static int g_live = 0;
static void test_grow() {
  auto alloc = [](void** np, size_t bytes) -> int {
    if (bytes > (1u << 20)) return fail(WDBX_E_NOMEM, "refused %zu bytes", bytes);
    *np = malloc(bytes);
    ++g_live;
    return *np ? WDBX_OK : WDBX_E_NOMEM;
  };
  auto release = [](void* p) -> int {
    free(p);
    --g_live;
    return WDBX_OK;
  };
  void* p = nullptr;
  size_t have = 0;
  CHECK(grow_with(&p, &have, 100, alloc, release) == WDBX_OK && p && have == 100);
  memset(p, 0xAB, 100);
  void* keep = p;
  CHECK(grow_with(&p, &have, 64, alloc, release) == WDBX_OK && p == keep && have == 100);  // large enough: untouched
  CHECK(grow_with(&p, &have, 4096, alloc, release) == WDBX_OK && have == 4096);
  memset(p, 0xCD, 4096);
  CHECK(grow_with(&p, &have, 2u << 20, alloc, release) == WDBX_E_NOMEM);                    // refused: the slot is EMPTY,
  CHECK(p == nullptr && have == 0 && g_live == 0);                                          // not dangling
  CHECK(g_err.find("refused") != std::string::npos);
  CHECK(grow_with(&p, &have, 10, alloc, release) == WDBX_OK && have == 10);                 // and usable again
  release(p);
  CHECK(g_live == 0);
}

// ---- 5. option table + the exception barrier of an extern "C" entry point ----
struct Handle {
  int64_t a = 1, b = 2, c = 3;
};
static const OptionDesc<Handle> kOpts[] = {{"alpha", &Handle::a}, {"beta", &Handle::b}, {"gamma", &Handle::c}};

extern "C" int entry_point_that_throws(int what) try {
  if (what == 0) return WDBX_OK;
  if (what == 1) throw std::bad_alloc();
  if (what == 2) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread start");
  if (what == 3) throw 42;
  std::vector<int> v;
  return v.at(10);  // std::out_of_range
} WDBX_CATCH

static void test_options_and_barrier() {
  Handle h;
  CHECK(find_option(&h, kOpts, "alpha") == &h.a && find_option(&h, kOpts, "gamma") == &h.c);
  CHECK(find_option(&h, kOpts, "delta") == nullptr && find_option(&h, kOpts, nullptr) == nullptr);
  *find_option(&h, kOpts, "beta") = 77;
  CHECK(h.b == 77);
  CHECK(entry_point_that_throws(0) == WDBX_OK);
  CHECK(entry_point_that_throws(1) == WDBX_E_NOMEM && g_err.find("bad_alloc") != std::string::npos);
  CHECK(entry_point_that_throws(2) == WDBX_E_STATE && g_err.find("thread start") != std::string::npos);
  CHECK(entry_point_that_throws(3) == WDBX_E_STATE && g_err.find("unknown exception") != std::string::npos);
  CHECK(entry_point_that_throws(4) == WDBX_E_STATE);
  // the message is per thread: another thread's failure does not overwrite this thread's
  const std::string mine = g_err;
  std::thread([] { (void)fail(WDBX_E_INVALID, "other thread"); }).join();
  CHECK(g_err == mine);
}

// ---- 6. two dispatchers driven from two threads at once (two groups in one process) ----
static void test_two_groups() {
  auto drive = [](int seed) {
    Dispatcher d;
    d.start({0, 1, 2, 3}, nullptr);
    long sum[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 2000; ++i) {
      const std::function<int(int)> job = [&](int s) -> int {
        sum[s] += i + seed;
        return (i % 97 == 0 && s == 2) ? fail(WDBX_E_RCCL, "collective %d", i) : WDBX_OK;
      };
      const int rc = d.run(job);
      CHECK(rc == ((i % 97 == 0) ? WDBX_E_RCCL : WDBX_OK));
    }
    for (int s = 1; s < 5; ++s) CHECK(sum[s] == sum[0]);
  };
  std::thread a(drive, 1), b(drive, 1000);
  a.join();
  b.join();
}

// ---- 7. the staging-slot pool: the call shape of search_host (wdbx_hip.hip) -- decide, take a slot or wait with the mutex
// released and decide again, change handle state and "enqueue" under the mutex, release the mutex, work on the slot,
// give it back; "masked" calls keep the mutex to their end; the owner is destroyed when every slot is free ----
struct FakeHandle {
  std::mutex mu;
  SlotPool<4> slots;
  int active_mask = 0;        // handle state a call sets for its own duration (guarded by mu)
  long enqueued = 0;          // "launches": guarded by mu
  long slot_data[4] = {0, 0, 0, 0};  // what a call reads/writes in its slot WITHOUT the mutex: exclusive by ownership
  long leaked_masks = 0;      // an unmasked call that saw another call's mask (the bug this shape once had)
};

static void slot_call(FakeHandle& h, bool masked, int work_us) {
  std::unique_lock<std::mutex> lk(h.mu);
  int slot = -1;
  for (;;) {  // take the slot FIRST, before touching handle state; decide again after every wait
    if ((slot = h.slots.try_take()) >= 0) break;
    h.slots.wait(lk);
  }
  struct Hold {
    FakeHandle& h;
    std::unique_lock<std::mutex>& lk;
    int slot;
    ~Hold() { h.slots.give_back(slot, lk); }
  } hold{h, lk, slot};
  if (!masked && h.active_mask) ++h.leaked_masks;
  if (masked) h.active_mask = 1;
  ++h.enqueued;
  const long mine = h.enqueued;
  if (!masked) lk.unlock();                      // the wait for the GPU: outside the mutex unless the call carries a mask
  h.slot_data[slot] = mine;                      // plain accesses: nobody else may own this slot now
  if (work_us) std::this_thread::sleep_for(std::chrono::microseconds(work_us));
  CHECK(h.slot_data[slot] == mine);
  if (masked) h.active_mask = 0;                 // (still under the mutex)
}

static void test_slot_pool() {
  FakeHandle h;
  std::vector<std::thread> ts;
  for (int t = 0; t < 10; ++t)
    ts.emplace_back([&h, t] {
      for (int i = 0; i < 1500; ++i) slot_call(h, (i + t) % 7 == 0, (i % 64 == 0) ? 50 : 0);
    });
  for (auto& th : ts) th.join();
  std::unique_lock<std::mutex> lk(h.mu);
  h.slots.wait_all_free(lk);                     // what the owner's destruction does
  CHECK(h.enqueued == 10L * 1500 && h.leaked_masks == 0 && h.active_mask == 0);
  for (bool b : h.slots.busy) CHECK(!b);
}

#ifdef HARNESS_PLANT_RACE
// negative control (-DHARNESS_PLANT_RACE): every shard's job bumps ONE plain counter -- the kind of unguarded shared state the
// reference has (indexing.py:381-383) and the dispatcher must not have.  ThreadSanitizer has to report it, or the clean run
// of the real tests above proves nothing.
static long g_planted = 0;
static void test_planted_race() {
  Dispatcher d;
  d.start({0, 1, 2, 3}, nullptr);
  const std::function<int(int)> job = [&](int) -> int {
    for (int i = 0; i < 1000; ++i) ++g_planted;
    return WDBX_OK;
  };
  for (int i = 0; i < 200; ++i) (void)d.run(job);
}
#endif

int main(int argc, char** argv) {
  const int dispatches = argc > 1 ? atoi(argv[1]) : 10000;
  const int workers = argc > 2 ? atoi(argv[2]) : 8;
#ifdef HARNESS_PLANT_RACE
  test_planted_race();
  printf("planted race ran to the end (%ld)\n", g_planted);
  return 0;
#endif
  test_dispatcher(dispatches, workers);
  test_start_stop_churn();
  test_ordered_locks();
  test_grow();
  test_options_and_barrier();
  test_two_groups();
  test_slot_pool();
  printf("harness ok: %d dispatches over %d workers, start/stop churn, ordered locks, grow, options, exception barrier, two groups, slot pool\n",
         dispatches, workers);
  return 0;
}
