"""Host-side mirror of kimera::SemanticLabel2Color (kimera_semantics/src/color.cpp:42-94):
the CSV -> {colour -> label, label -> colour} maps that configure the integrator.  Runs once
at start-up (not on the hot path); its output feeds ks_config.label_rgba and
ks_set_color_to_label.  Quirks of the reference loader are kept on purpose:
  * the header line is parsed as a data row (atoi("red") == 0) -> colour (0,0,0,0) maps to id 0;
  * later rows overwrite earlier ones in both maps (color.cpp:58-59);
  * id 0 is finally forced to White and White to id 0 (color.cpp:64-66);
  * lookups use alpha 255 (semantic_tsdf_integrator_fast.cpp:157) and compare all four
    channels (color.cpp:25-27): keys whose CSV alpha is not 255 can never match;
  * unknown colour -> label 0 (color.cpp:72-81); unknown label -> colour (0,0,0,0) (color.cpp:86-93).
"""
from __future__ import annotations

import numpy as np


def _atoi(s: str) -> int:
    s = s.strip()
    sign, i = 1, 0
    if s[:1] in "+-":
        sign = -1 if s[0] == "-" else 1
        i = 1
    j = i
    while j < len(s) and s[j].isdigit():
        j += 1
    return sign * int(s[i:j]) if j > i else 0


class SemanticLabel2Color:
    def __init__(self, filename: str):
        self.color_to_semantic_label: dict[tuple, int] = {}
        self.semantic_label_to_color_map: dict[int, tuple] = {}
        with open(filename) as fh:
            for row_number, line in enumerate(fh, 1):
                line = line.rstrip("\r\n")
                if not line:
                    continue
                cells = line.split(",")
                if len(cells) != 6:
                    raise ValueError(f"Row {row_number} is invalid.")  # CHECK_EQ(loop->size(), 6)
                r, g, b, a, idv = (_atoi(c) & 0xFF for c in cells[1:6])
                self.semantic_label_to_color_map[idv] = (r, g, b, a)
                self.color_to_semantic_label[(r, g, b, a)] = idv
        self.semantic_label_to_color_map[0] = (255, 255, 255, 255)
        self.color_to_semantic_label[(255, 255, 255, 255)] = 0

    def get_semantic_label_from_color(self, rgba) -> int:
        return self.color_to_semantic_label.get(tuple(int(x) for x in rgba), 0)

    def get_color_from_semantic_label(self, label: int):
        return self.semantic_label_to_color_map.get(int(label), (0, 0, 0, 0))

    # ---- what the C ABI consumes ----
    def label_rgba_table(self) -> np.ndarray:
        t = np.zeros((256, 4), dtype=np.uint8)
        for k, v in self.semantic_label_to_color_map.items():
            t[k] = v
        return t

    def color_keys(self):
        keys = np.array(list(self.color_to_semantic_label.keys()), dtype=np.uint8).reshape(-1, 4)
        labels = np.array(list(self.color_to_semantic_label.values()), dtype=np.uint8)
        return keys, labels
