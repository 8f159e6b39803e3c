"""Drum pitch maps of the reference (``utils/mapping_utils.py``), as data.

  GM_TO_CUSTOM      General-MIDI percussion key 35..81 -> the 26 "custom GM" pitches 35..60
                    the model is trained on (mapping_utils.py:3-51)
  ADTOF_MAPPING     custom pitch 35..61 -> 8 ADTOF classes (mapping_utils.py:57-85)
  ADTOF_INVERSE_MAPPING  class -> member custom pitches (mapping_utils.py:87-96)
  ADTOF_LABEL       class pitch -> instrument label (mapping_utils.py:98-107)
"""

def _expand(groups):
    return {src: dst for dst, srcs in groups.items() for src in srcs}


GM_TO_CUSTOM = _expand({
    35: [35], 36: [36], 37: [37], 38: [38], 39: [39], 40: [40], 41: [41, 43, 45], 42: [42], 43: [44], 44: [46],
    45: [47, 48], 46: [49, 57], 47: [50], 48: [51, 53, 59], 49: [52], 50: [54], 51: [55], 52: [56, 67, 68], 53: [58],
    54: [60, 61, 62, 63, 64, 65, 66], 55: [69, 70], 56: [71, 72], 57: [73, 74], 58: [75, 76, 77], 59: [78, 79],
    60: [80, 81],
})

ADTOF_INVERSE_MAPPING = {
    35: [35, 36], 38: [37, 38, 39, 40], 41: [41, 45, 47], 42: [42, 43, 44, 50], 48: [46, 48, 49, 51], 52: [52],
    58: [58], 61: [53, 54, 55, 56, 57, 59, 60],
}
ADTOF_MAPPING = _expand(ADTOF_INVERSE_MAPPING)
ADTOF_MAPPING[61] = 61

ADTOF_LABEL = {35: "BD", 38: "SD", 41: "TT", 42: "HH", 48: "CY + RD", 52: "Cowbell", 58: "Claves", 61: "Other"}
ADTOF_LABEL_TO_PITCH = {v: k for k, v in ADTOF_LABEL.items()}
