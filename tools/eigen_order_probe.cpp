// Which fp32 summation order does THIS Eigen give `Isometry3f * Vector4f` (fast_apdgicp_impl.hpp:149)?  Compile against the Eigen
// the reference is built with, with the reference's flags:   g++ -O2 -msse4.2 -I/usr/include/eigen3 eigen_order_probe.cpp && ./a.out
// prints "pairwise" (Eigen >= 3.3: the library's default) or "linear chain" (Eigen 3.2: setTransformOrder(EIGEN_LINEAR_CHAIN) /
// APDGICP_FLAG_XF_LINEAR_CHAIN), or the counts when neither formula reproduces the product (then tell us: parity is by tolerance only).
#include <Eigen/Geometry>
#include <cstdio>
#include <cstring>
#include <random>
int main() {
  std::mt19937 rng(20241022);
  std::uniform_real_distribution<float> U(-100.f, 100.f), A(-3.2f, 3.2f);
  long pairwise = 0, chain = 0, telling = 0;
  for (int it = 0; it < 2000; it++) {
    Eigen::Isometry3f T = Eigen::Isometry3f::Identity();
    T.linear() = (Eigen::AngleAxisf(A(rng), Eigen::Vector3f::UnitZ()) * Eigen::AngleAxisf(0.1f * A(rng), Eigen::Vector3f::UnitY())).toRotationMatrix();
    T.translation() = 0.03f * Eigen::Vector3f(U(rng), U(rng), U(rng));
    alignas(16) float p[4] = {U(rng), U(rng), 0.1f * U(rng), 1.f}, out[4];   // pcl::PointXYZI::data: {x, y, z, 1}
    Eigen::Map<Eigen::Vector4f, Eigen::Aligned>(out) = T * Eigen::Map<const Eigen::Vector4f, Eigen::Aligned>(p);   // getVector4fMap()
    for (int r = 0; r < 3; r++) {
      volatile float a = T(r, 0) * p[0], b = T(r, 1) * p[1], c = T(r, 2) * p[2], t = T(r, 3);   // (volatile: no contraction, no reassociation)
      volatile float ab = a + b, ct = c + t, abc = ab + c;
      const float vp = ab + ct, vc = abc + t;
      if (std::memcmp(&vp, &vc, 4) == 0) continue;   // both formulas agree on this input: not telling
      telling++, pairwise += std::memcmp(&out[r], &vp, 4) == 0, chain += std::memcmp(&out[r], &vc, 4) == 0;
    }
  }
  std::printf("Eigen %d.%d.%d: %ld telling coefficients, pairwise matches %ld, linear chain matches %ld -> %s\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION,
              EIGEN_MINOR_VERSION, telling, pairwise, chain, pairwise == telling ? "pairwise" : chain == telling ? "linear chain" : "NEITHER");
  return pairwise == telling || chain == telling ? 0 : 1;
}
