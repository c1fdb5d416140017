// Stand-in for voxblox/core/color.h: RGBA colour, named colours, blending, rainbow map.
#pragma once
#include <cmath>
#include <cstdint>

#include "voxblox/core/common.h"

namespace voxblox {

struct Color {
  Color() : r(0), g(0), b(0), a(0) {}
  Color(uint8_t _r, uint8_t _g, uint8_t _b) : Color(_r, _g, _b, 255) {}
  Color(uint8_t _r, uint8_t _g, uint8_t _b, uint8_t _a) : r(_r), g(_g), b(_b), a(_a) {}
  uint8_t r, g, b, a;

  static Color blendTwoColors(const Color& first_color, FloatingPoint first_weight, const Color& second_color,
                              FloatingPoint second_weight) {
    FloatingPoint total_weight = first_weight + second_weight;
    first_weight /= total_weight;
    second_weight /= total_weight;
    Color new_color;
    new_color.r = static_cast<uint8_t>(round(first_color.r * first_weight + second_color.r * second_weight));
    new_color.g = static_cast<uint8_t>(round(first_color.g * first_weight + second_color.g * second_weight));
    new_color.b = static_cast<uint8_t>(round(first_color.b * first_weight + second_color.b * second_weight));
    new_color.a = static_cast<uint8_t>(round(first_color.a * first_weight + second_color.a * second_weight));
    return new_color;
  }
  static const Color White() { return Color(255, 255, 255); }
  static const Color Black() { return Color(0, 0, 0); }
  static const Color Gray() { return Color(127, 127, 127); }
  static const Color Red() { return Color(255, 0, 0); }
  static const Color Green() { return Color(0, 255, 0); }
  static const Color Blue() { return Color(0, 0, 255); }
  static const Color Yellow() { return Color(255, 255, 0); }
  static const Color Orange() { return Color(255, 127, 0); }
  static const Color Purple() { return Color(127, 0, 255); }
  static const Color Teal() { return Color(0, 255, 255); }
  static const Color Pink() { return Color(255, 0, 127); }
};
typedef AlignedVector<Color> Colors;

inline Color rainbowColorMap(double h) {
  Color color;
  color.a = 255;
  double s = 1.0;
  double v = 1.0;
  h -= floor(h);
  h *= 6;
  int i;
  double m, n, f;
  i = floor(h);
  f = h - i;
  if (!(i & 1)) f = 1 - f;
  m = v * (1 - s);
  n = v * (1 - s * f);
  switch (i) {
    case 6:
    case 0: color.r = 255 * v; color.g = 255 * n; color.b = 255 * m; break;
    case 1: color.r = 255 * n; color.g = 255 * v; color.b = 255 * m; break;
    case 2: color.r = 255 * m; color.g = 255 * v; color.b = 255 * n; break;
    case 3: color.r = 255 * m; color.g = 255 * n; color.b = 255 * v; break;
    case 4: color.r = 255 * n; color.g = 255 * m; color.b = 255 * v; break;
    case 5: color.r = 255 * v; color.g = 255 * m; color.b = 255 * n; break;
    default: color.r = 255; color.g = 127; color.b = 127; break;
  }
  return color;
}

inline Color randomColor() { return Color(rand() % 256, rand() % 256, rand() % 256); }

}  // namespace voxblox
