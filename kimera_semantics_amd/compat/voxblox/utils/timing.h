// Stand-in for voxblox/utils/timing.h: named scope timers (accumulated, printable).
#pragma once
#include <chrono>
#include <map>
#include <mutex>
#include <sstream>
#include <string>

namespace voxblox {
namespace timing {
class Timing {
 public:
  static std::map<std::string, std::pair<double, size_t>>& table() {
    static std::map<std::string, std::pair<double, size_t>> t;
    return t;
  }
  static std::mutex& mutex() {
    static std::mutex m;
    return m;
  }
  static std::string Print() {
    std::lock_guard<std::mutex> l(mutex());
    std::ostringstream os;
    for (auto& kv : table()) os << kv.first << "\t" << kv.second.second << "\t" << kv.second.first << " s\n";
    return os.str();
  }
  static void Reset() {
    std::lock_guard<std::mutex> l(mutex());
    table().clear();
  }
};
class Timer {
 public:
  explicit Timer(const std::string& tag, bool construct_stopped = false) : tag_(tag), timing_(false) {
    if (!construct_stopped) Start();
  }
  ~Timer() {
    if (timing_) Stop();
  }
  void Start() {
    timing_ = true;
    t0_ = std::chrono::steady_clock::now();
  }
  void Stop() {
    if (!timing_) return;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
    timing_ = false;
    std::lock_guard<std::mutex> l(Timing::mutex());
    auto& e = Timing::table()[tag_];
    e.first += dt;
    e.second += 1;
  }

 private:
  std::string tag_;
  bool timing_;
  std::chrono::steady_clock::time_point t0_;
};
}  // namespace timing
}  // namespace voxblox
