// Host-side matrix helper shared by the C-ABI module and the camera code.
#pragma once

namespace rt {

// General 4x4 inverse by 2x2 sub-determinants, column-major, plain fp32 with one
// rounding per operator.  Stands in for the GLSL inverse() the shader applies to
// ubo.projection and ubo.view per sample (ray_tracing.comp:230,233).
inline void mat4_inverse(const float *m, float *out) {
  auto A = [m](int r, int c) { return m[c * 4 + r]; };
  float s0 = A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1);
  float s1 = A(0, 0) * A(1, 2) - A(1, 0) * A(0, 2);
  float s2 = A(0, 0) * A(1, 3) - A(1, 0) * A(0, 3);
  float s3 = A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2);
  float s4 = A(0, 1) * A(1, 3) - A(1, 1) * A(0, 3);
  float s5 = A(0, 2) * A(1, 3) - A(1, 2) * A(0, 3);
  float c5 = A(2, 2) * A(3, 3) - A(3, 2) * A(2, 3);
  float c4 = A(2, 1) * A(3, 3) - A(3, 1) * A(2, 3);
  float c3 = A(2, 1) * A(3, 2) - A(3, 1) * A(2, 2);
  float c2 = A(2, 0) * A(3, 3) - A(3, 0) * A(2, 3);
  float c1 = A(2, 0) * A(3, 2) - A(3, 0) * A(2, 2);
  float c0 = A(2, 0) * A(3, 1) - A(3, 0) * A(2, 1);
  float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  float id = 1.0f / det;
  float b[4][4];
  b[0][0] = (A(1, 1) * c5 - A(1, 2) * c4 + A(1, 3) * c3) * id;
  b[0][1] = (-A(0, 1) * c5 + A(0, 2) * c4 - A(0, 3) * c3) * id;
  b[0][2] = (A(3, 1) * s5 - A(3, 2) * s4 + A(3, 3) * s3) * id;
  b[0][3] = (-A(2, 1) * s5 + A(2, 2) * s4 - A(2, 3) * s3) * id;
  b[1][0] = (-A(1, 0) * c5 + A(1, 2) * c2 - A(1, 3) * c1) * id;
  b[1][1] = (A(0, 0) * c5 - A(0, 2) * c2 + A(0, 3) * c1) * id;
  b[1][2] = (-A(3, 0) * s5 + A(3, 2) * s2 - A(3, 3) * s1) * id;
  b[1][3] = (A(2, 0) * s5 - A(2, 2) * s2 + A(2, 3) * s1) * id;
  b[2][0] = (A(1, 0) * c4 - A(1, 1) * c2 + A(1, 3) * c0) * id;
  b[2][1] = (-A(0, 0) * c4 + A(0, 1) * c2 - A(0, 3) * c0) * id;
  b[2][2] = (A(3, 0) * s4 - A(3, 1) * s2 + A(3, 3) * s0) * id;
  b[2][3] = (-A(2, 0) * s4 + A(2, 1) * s2 - A(2, 3) * s0) * id;
  b[3][0] = (-A(1, 0) * c3 + A(1, 1) * c1 - A(1, 2) * c0) * id;
  b[3][1] = (A(0, 0) * c3 - A(0, 1) * c1 + A(0, 2) * c0) * id;
  b[3][2] = (-A(3, 0) * s3 + A(3, 1) * s1 - A(3, 2) * s0) * id;
  b[3][3] = (A(2, 0) * s3 - A(2, 1) * s1 + A(2, 2) * s0) * id;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) out[c * 4 + r] = b[r][c];
}

}  // namespace rt
