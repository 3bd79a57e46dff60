// core/interval.hpp -- rt::Interval (reference: include/core/interval.cuh:4-48)
#pragma once

namespace rt {

class Interval {
public:
  Interval(float lower, float upper, bool lower_closed = true, bool upper_closed = true)
    : lower_(lower), upper_(upper), lower_closed_(lower_closed), upper_closed_(upper_closed)
  {}

  bool contains(float f) const
  {
    const bool lo = lower_closed_ ? (lower_ <= f) : (lower_ < f);
    const bool hi = upper_closed_ ? (f <= upper_) : (f < upper_);
    return lo && hi;
  }
  float lower() const { return lower_; }
  float upper() const { return upper_; }
  // (the render path only ever uses the closed/closed mode, src/main.cu:60 and core/light.cuh:31; rt::hit() takes all four)
  bool lower_closed() const { return lower_closed_; }
  bool upper_closed() const { return upper_closed_; }
  bool is_closed() const { return lower_closed_ && upper_closed_; }

private:
  float lower_, upper_;
  bool lower_closed_, upper_closed_;
};

} // namespace rt
