/*
 * noise_oracle.c — CPU restatement of soil::noise (source/soillib/op/noise.hpp:14-56):
 * 3-D OpenSimplex2 noise (K.jpg's published algorithm as shipped in the
 * third-party FastNoiseLite, vendored by the reference under
 * source/soillib/external/FastNoiseLite.h), summed as FBm.
 * TEST INFRASTRUCTURE ONLY.  This function IS pinned: tests compare it bit for
 * bit with oracle/_ref/libfnl_ref.so (the vendored header compiled in place)
 * and with the committed fixtures tests/golden/noise_*.npy produced by it.
 *
 * FastNoiseLite.h line references below are to that vendored header.
 */
#include <math.h>
#include <stdint.h>

#include "soil_oracle.h"

/* hashing primes, FastNoiseLite.h:487-489 */
#define ORC_PX 501125321
#define ORC_PY 1136930381
#define ORC_PZ 1720413743

/* The 64 gradients of Lookup::Gradients3D (FastNoiseLite.h:2529-2547): the 12
 * cube-edge directions, five times over, then four fill-ins. */
static void orc_grad3(int idx, float g[3]) {
  static const signed char fill[4][3] = {{1, 1, 0}, {0, -1, 1}, {-1, 1, 0}, {0, -1, -1}};
  if (idx >= 60) {
    g[0] = fill[idx - 60][0];
    g[1] = fill[idx - 60][1];
    g[2] = fill[idx - 60][2];
    return;
  }
  const int e = idx % 12, zero_axis = e / 4, s = e % 4;
  const float a = (s & 1) ? -1.0f : 1.0f; /* first non-zero component */
  const float b = (s & 2) ? -1.0f : 1.0f; /* second non-zero component */
  if (zero_axis == 0) { g[0] = 0; g[1] = a; g[2] = b; }
  else if (zero_axis == 1) { g[0] = a; g[1] = 0; g[2] = b; }
  else { g[0] = a; g[1] = b; g[2] = 0; }
}

/* Hash + GradCoord, FastNoiseLite.h:500-506, :542-553 (int arithmetic wraps) */
static float orc_grad_coord(int32_t seed, int32_t xp, int32_t yp, int32_t zp, float xd, float yd,
                            float zd) {
  uint32_t h = (uint32_t)(seed ^ xp ^ yp ^ zp);
  h *= 0x27d4eb2du;
  int32_t hs = (int32_t)h;
  hs ^= hs >> 15; /* arithmetic shift, as on every supported compiler */
  hs &= 63 << 2;
  float g[3];
  orc_grad3(hs >> 2, g);
  return xd * g[0] + yd * g[1] + zd * g[2];
}

static int32_t orc_mul_wrap(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static int32_t orc_fast_round(float f) { /* :453 */
  return f >= 0 ? (int32_t)(f + 0.5f) : (int32_t)(f - 0.5f);
}

/* SingleOpenSimplex2 (3-D), FastNoiseLite.h:1054-1150 */
static float orc_opensimplex2_3d(int32_t seed, float x, float y, float z) {
  int32_t i = orc_fast_round(x), j = orc_fast_round(y), k = orc_fast_round(z); /* :1065-1067 */
  float x0 = (float)(x - i), y0 = (float)(y - j), z0 = (float)(z - k);        /* :1068-1070 */
  int32_t xs = (int32_t)(-1.0f - x0) | 1;                                      /* :1072-1074 */
  int32_t ys = (int32_t)(-1.0f - y0) | 1;
  int32_t zs = (int32_t)(-1.0f - z0) | 1;
  float ax0 = xs * -x0, ay0 = ys * -y0, az0 = zs * -z0; /* :1076-1078 */
  i = orc_mul_wrap(i, ORC_PX);                          /* :1080-1082 */
  j = orc_mul_wrap(j, ORC_PY);
  k = orc_mul_wrap(k, ORC_PZ);
  float value = 0;
  float a = (0.6f - x0 * x0) - (y0 * y0 + z0 * z0); /* :1085 */
  for (int l = 0;; l++) {
    if (a > 0) value += (a * a) * (a * a) * orc_grad_coord(seed, i, j, k, x0, y0, z0); /* :1089-1092 */
    float b = a + 1; /* :1094 */
    int32_t i1 = i, j1 = j, k1 = k;
    float x1 = x0, y1 = y0, z1 = z0;
    if (ax0 >= ay0 && ax0 >= az0) { /* :1102-1107 */
      x1 += xs;
      b -= xs * 2 * x1;
      i1 -= orc_mul_wrap(xs, ORC_PX);
    } else if (ay0 > ax0 && ay0 >= az0) { /* :1108-1113 */
      y1 += ys;
      b -= ys * 2 * y1;
      j1 -= orc_mul_wrap(ys, ORC_PY);
    } else { /* :1114-1119 */
      z1 += zs;
      b -= zs * 2 * z1;
      k1 -= orc_mul_wrap(zs, ORC_PZ);
    }
    if (b > 0) value += (b * b) * (b * b) * orc_grad_coord(seed, i1, j1, k1, x1, y1, z1); /* :1121-1124 */
    if (l == 1) break;                                                                    /* :1126 */
    ax0 = 0.5f - ax0; /* :1128-1130 */
    ay0 = 0.5f - ay0;
    az0 = 0.5f - az0;
    x0 = xs * ax0; /* :1132-1134 */
    y0 = ys * ay0;
    z0 = zs * az0;
    a += (0.75f - ax0) - (ay0 + az0); /* :1136 */
    i += (xs >> 1) & ORC_PX;          /* :1138-1140 */
    j += (ys >> 1) & ORC_PY;
    k += (zs >> 1) & ORC_PZ;
    xs = -xs; /* :1142-1144 */
    ys = -ys;
    zs = -zs;
    seed = ~seed; /* :1146 */
  }
  return value * 32.69428253173828125f; /* :1149 */
}

/* soil::noise, noise.hpp:42-56 with noise_param_t::update :16-23 and operator() :37-39 */
void orc_noise(float* out, int64_t H, int64_t W, const orc_noise_param* p) {
  /* CalculateFractalBounding, FastNoiseLite.h:473-484 */
  const float gain = p->gain < 0 ? -p->gain : p->gain;
  float amp0 = gain, ampFractal = 1.0f;
  for (int o = 1; o < p->octaves; o++) {
    ampFractal += amp0;
    amp0 *= gain;
  }
  const float bounding = 1 / ampFractal;
  const float weighted = 0.0f; /* mWeightedStrength default, :126 */

  for (int64_t n = 0; n < H * W; ++n) {
    const int32_t px = (int32_t)(n / W), py = (int32_t)(n % W);
    float x = px / p->ext[0], y = py / p->ext[1], z = p->seed; /* noise.hpp:38 */
    x *= p->frequency; /* TransformNoiseCoordinate, FastNoiseLite.h:689-691 */
    y *= p->frequency;
    z *= p->frequency;
    { /* TransformType3D_DefaultOpenSimplex2, :715-722 */
      const float R3 = (float)(2.0 / 3.0);
      const float r = (x + y + z) * R3;
      x = r - x;
      y = r - y;
      z = r - z;
    }
    int32_t seed = 1337; /* FastNoiseLite default seed, :114 (noise_param_t never sets it) */
    float sum = 0, amp = bounding; /* GenFractalFBm, :866-885 */
    for (int o = 0; o < p->octaves; o++) {
      const float noise = orc_opensimplex2_3d(seed++, x, y, z);
      sum += noise * amp;
      amp *= 1.0f + weighted * ((noise + 1) * 0.5f - 1.0f); /* Lerp(1, (noise+1)/2, w), :456, :876 */
      x *= p->lacunarity;
      y *= p->lacunarity;
      z *= p->lacunarity;
      amp *= p->gain;
    }
    out[n] = sum;
  }
}
