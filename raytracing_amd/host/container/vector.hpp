// container/vector.hpp -- rt::Vector<T>, rt::VectorView<T>, rt::make_view
// (reference API: include/container/vector.cuh:15-256).
// The reference keeps scene data in CUDA managed memory and hands kernels a view.  Here
// the scene is flattened to SoA and copied to HBM at upload time (pipeline/render.hpp),
// so the host container is plain host memory with the same surface.
#pragma once
#include <cassert>
#include <cstddef>
#include <initializer_list>
#include <vector>

namespace rt {

template <typename T> class VectorView {
public:
  VectorView(const T *data, std::size_t size) : data_(data), size_(size) {}
  std::size_t size() const { return size_; }
  const T &operator[](std::size_t i) const { assert(i < size_); return data_[i]; }
  const T *begin() const { return data_; }
  const T *end() const { return data_ + size_; }

private:
  const T *data_;
  std::size_t size_;
};

template <typename T> class Vector {
public:
  Vector() = default;
  Vector(std::initializer_list<T> init) : v_(init) {}
  explicit Vector(std::vector<T> v) : v_(std::move(v)) {}
  Vector(const Vector &) = delete; // move-only owner, like the reference
  Vector &operator=(const Vector &) = delete;
  Vector(Vector &&) = default;
  Vector &operator=(Vector &&) = default;

  void push_back(const T &x) { v_.push_back(x); }
  template <typename... Args> T &emplace_back(Args &&...args) { return v_.emplace_back(std::forward<Args>(args)...); }
  std::size_t size() const { return v_.size(); }
  bool empty() const { return v_.empty(); }
  T &operator[](std::size_t i) { assert(i < v_.size()); return v_[i]; }
  const T &operator[](std::size_t i) const { assert(i < v_.size()); return v_[i]; }
  T *begin() { return v_.data(); }
  T *end() { return v_.data() + v_.size(); }
  const T *begin() const { return v_.data(); }
  const T *end() const { return v_.data() + v_.size(); }
  const T *data() const { return v_.data(); }

private:
  std::vector<T> v_;
};

template <typename T> VectorView<T> make_view(const Vector<T> &v) { return VectorView<T>(v.data(), v.size()); }

} // namespace rt
