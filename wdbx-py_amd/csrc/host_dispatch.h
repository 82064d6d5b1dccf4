// host_dispatch.h -- the part of the host side that needs NO device: error plumbing and the exception barrier of the C ABI,
// the per-shard worker threads and their dispatcher, ordered multi-handle locking, grow-and-free buffer bookkeeping, the
// option table lookup.  Included by wdbx_hip.hip (before the kernels) -- and, on its own, by tests/host_harness/
// dispatch_harness.cpp, which plain g++ builds with -fsanitize=thread and -fsanitize=address,undefined in the CPU suite
// (tests/test_host_dispatch_sanitizers.py): SURVEY section 5 asks for the host side under TSAN because the reference mutates
// shared dicts from pool workers with no locks at all (wdbx/core/indexing.py:381-383 under run_in_executor :407).
// No HIP, no RCCL, no kernel types in here: whatever touches a device is passed in as a function.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#ifndef WDBX_OK  // (the harness includes this header without include/wdbx_hip.h)
#define WDBX_OK 0
#define WDBX_E_INVALID (-1)
#define WDBX_E_HIP (-2)
#define WDBX_E_NOMEM (-3)
#define WDBX_E_NODEVICE (-4)
#define WDBX_E_RCCL (-5)
#define WDBX_E_STATE (-6)
#endif

// ------------------------------------------------------------------------------------------------
// error plumbing: one message per calling thread (wdbx_last_error)
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) noexcept {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  try {
    g_err = buf;
  } catch (...) {  // (the message is lost, the code is not)
  }
  return code;
}

// The exception barrier of the C ABI ("never throws", include/wdbx_hip.h): every extern "C" entry point is a
// function-try-block ending in one of these handlers, so nothing the host side throws (std::bad_alloc from a vector or
// string, std::system_error from a mutex or a thread) can unwind into the caller's ctypes / cgo / JNI frame, where it
// would be std::terminate.  The reference's convention for backend failures is "log and return []"
// (wdbx/core/indexing.py:1028-1030), never a dead interpreter.
#define WDBX_CATCH                                                                                   \
  catch (const std::bad_alloc&) { return fail(WDBX_E_NOMEM, "host allocation failed (std::bad_alloc)"); } \
  catch (const std::exception& e_) { return fail(WDBX_E_STATE, "internal error: %s", e_.what()); }   \
  catch (...) { return fail(WDBX_E_STATE, "internal error: unknown exception"); }
#define WDBX_CATCH_VOID                                                  \
  catch (const std::exception& e_) { (void)fail(WDBX_E_STATE, "internal error: %s", e_.what()); } \
  catch (...) { (void)fail(WDBX_E_STATE, "internal error: unknown exception"); }

// ------------------------------------------------------------------------------------------------
// grow-and-free bookkeeping of the scratch buffers: *p holds *have bytes; a larger need frees and re-allocates (contents are
// scratch).  alloc / release return 0 on success.  After a failed allocation the slot is EMPTY (null, 0), never dangling.
// ------------------------------------------------------------------------------------------------
template <class Alloc, class Release>
static int grow_with(void** p, size_t* have, size_t need, Alloc alloc, Release release) {
  if (need <= *have) return WDBX_OK;
  if (*p) {
    const int rc = release(*p);
    *p = nullptr;
    *have = 0;
    if (rc) return rc;
  }
  *p = nullptr;
  *have = 0;
  void* np = nullptr;
  const int rc = alloc(&np, need);
  if (rc) return rc;
  *p = np;
  *have = need;
  return WDBX_OK;
}

// ------------------------------------------------------------------------------------------------
// several handle mutexes at once, always in the order given (the group passes its shards in shard order and is the only
// multi-handle locker, so two group calls -- or a group call and a per-handle call -- can never hold them crosswise)
// ------------------------------------------------------------------------------------------------
struct OrderedLocks {
  std::vector<std::unique_lock<std::mutex>> held;
  OrderedLocks() = default;
  explicit OrderedLocks(const std::vector<std::mutex*>& mus) {
    held.reserve(mus.size());
    for (std::mutex* m : mus) held.emplace_back(*m);
  }
};

// ------------------------------------------------------------------------------------------------
// A pool of N staging slots guarded by the OWNER's mutex (the handle's): a small blocking search takes one, enqueues its
// launches, RELEASES the mutex while it waits for the GPU and reads its results from the slot, then gives the slot back.
// Every method is called with the owner's mutex held (through `lk`); wait() releases it while it sleeps.
//   * try_take(): a free slot or -1 -- never waits, so the caller may take it in the middle of its decision making;
//   * wait(lk): sleeps until some slot is given back.  The caller must then look at everything it decided on again: the
//     mutex was released, other calls ran (a call that had already changed handle state before waiting -- a row mask, say --
//     would have leaked it into them);
//   * give_back(slot, lk): with or without the mutex held on entry; leaves the mutex as it found it;
//   * wait_all_free(lk): for the owner's destruction.
// ------------------------------------------------------------------------------------------------
template <int N>
struct SlotPool {
  bool busy[N] = {};
  std::condition_variable cv;
  int try_take() {
    for (int s = 0; s < N; ++s)
      if (!busy[s]) {
        busy[s] = true;
        return s;
      }
    return -1;
  }
  void wait(std::unique_lock<std::mutex>& lk) { cv.wait(lk); }
  void give_back(int slot, std::unique_lock<std::mutex>& lk) {
    const bool had = lk.owns_lock();
    if (!had) lk.lock();
    busy[slot] = false;
    if (!had) lk.unlock();
    cv.notify_all();  // (waiters for "any slot" and a waiter for "all slots" share the condition variable)
  }
  void wait_all_free(std::unique_lock<std::mutex>& lk) {
    cv.wait(lk, [&] {
      for (bool b : busy)
        if (b) return false;
      return true;
    });
  }
};

// ------------------------------------------------------------------------------------------------
// option table: name -> int64 slot of a handle
// ------------------------------------------------------------------------------------------------
template <class T>
struct OptionDesc {
  const char* name;
  int64_t T::*slot;
};

template <class T, size_t N>
static int64_t* find_option(T* obj, const OptionDesc<T> (&table)[N], const char* name) {
  if (!name) return nullptr;
  for (const OptionDesc<T>& d : table)
    if (!strcmp(d.name, name)) return &(obj->*(d.slot));
  return nullptr;
}

// ------------------------------------------------------------------------------------------------
// The dispatcher of the in-process shard group: one persistent host thread per shard 1 .. S-1, bound to its shard's device at
// start (bind); shard 0 runs on the calling thread.  run(job) hands ONE job to all of them at once and returns when every
// shard has finished it: S chains of launches are enqueued in parallel instead of one thread walking S devices.
//   * hand-over: the job pointer is written before `posted` is released and read after it is acquired; rc / err are written
//     before `done` is released and read after it is acquired -- no other shared state;
//   * an idle worker spins ~200 us (a stream of lone queries finds it awake), then sleeps on its condition variable; the
//     dispatcher posts under the worker's mutex, so the wake-up cannot be lost;
//   * a job that fails or THROWS on a worker is reported as that shard's error code / message; nothing leaves run() while a
//     worker still uses the caller's job object.
// ------------------------------------------------------------------------------------------------
struct DispatchWorker {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::atomic<uint64_t> posted{0}, done{0};
  const std::function<int(int)>* job = nullptr;
  int shard = 0, device = 0, rc = 0;
  std::string err;
  bool stop = false;
  void (*bind)(int device) = nullptr;
};

static void dispatch_worker_main(DispatchWorker* w) {
  if (w->bind) w->bind(w->device);
  uint64_t seen = 0;
  for (;;) {
    uint64_t p = w->posted.load(std::memory_order_acquire);
    if (p == seen) {
      // ~200 us of spinning: a stream of lone queries (80-100 us each on a 1.25 M-row shard) finds the worker awake
      const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
      while (p == seen && std::chrono::steady_clock::now() < until) p = w->posted.load(std::memory_order_acquire);
      if (p == seen) {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->stop || w->posted.load(std::memory_order_acquire) != seen; });
        if (w->stop) return;
        p = w->posted.load(std::memory_order_acquire);
      }
    }
    {
      std::lock_guard<std::mutex> lk(w->m);
      if (w->stop) return;
    }
    int rc;
    try {
      rc = (*w->job)(w->shard);
    } catch (const std::exception& e) {
      rc = fail(WDBX_E_STATE, "internal error in shard %d's worker: %s", w->shard, e.what());
    } catch (...) {
      rc = fail(WDBX_E_STATE, "internal error in shard %d's worker", w->shard);
    }
    w->rc = rc;
    if (rc) w->err = g_err;
    seen = p;
    w->done.store(p, std::memory_order_release);
  }
}

struct Dispatcher {
  std::vector<std::unique_ptr<DispatchWorker>> workers;  // shards 1 .. S-1
  uint64_t dispatches = 0;

  // one worker per entry of `devices` (shard i + 1 on devices[i]); bind(device) runs first on the new thread
  void start(const std::vector<int>& devices, void (*bind)(int device)) {
    for (size_t i = 0; i < devices.size(); ++i) {
      std::unique_ptr<DispatchWorker> w(new DispatchWorker());
      w->shard = (int)i + 1;
      w->device = devices[i];
      w->bind = bind;
      w->th = std::thread(dispatch_worker_main, w.get());
      workers.push_back(std::move(w));
    }
  }

  // job(s) for every shard s at once: shard 0 here, the others on their workers.  Returns the first failure in shard order
  // (its message becomes the caller's wdbx_last_error).  One run() at a time (the group's mutex).
  int run(const std::function<int(int)>& job) {
    const uint64_t seq = ++dispatches;
    for (auto& w : workers) {
      w->job = &job;
      {
        std::lock_guard<std::mutex> lk(w->m);  // (pairs with the worker's wait: no lost wake-up)
        w->posted.store(seq, std::memory_order_release);
      }
      w->cv.notify_one();
    }
    int rc0;
    try {  // (nothing may leave this function while a worker still runs the caller's job object)
      rc0 = job(0);
    } catch (const std::exception& e) {
      rc0 = fail(WDBX_E_STATE, "internal error in shard 0's job: %s", e.what());
    } catch (...) {
      rc0 = fail(WDBX_E_STATE, "internal error in shard 0's job");
    }
    const std::string err0 = rc0 ? g_err : std::string();
    int rc = rc0;
    for (auto& w : workers) {
      int spins = 0;
      while (w->done.load(std::memory_order_acquire) != seq)
        if (++spins > 4000) std::this_thread::yield();
      if (rc == WDBX_OK && w->rc != WDBX_OK) {
        rc = w->rc;
        g_err = w->err;
      }
    }
    if (rc0) g_err = err0;
    return rc;
  }

  void stop() {
    for (auto& w : workers) {
      {
        std::lock_guard<std::mutex> lk(w->m);
        w->stop = true;
      }
      w->cv.notify_one();
      if (w->th.joinable()) w->th.join();
    }
    workers.clear();
  }

  ~Dispatcher() { stop(); }
};
