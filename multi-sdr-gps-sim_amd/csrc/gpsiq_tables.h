// gpsiq_tables.h — constant data of the GPS L1 C/A signal used by libgpsiq (host side).
//
// carrier LUT: the reference's sinTable512/cosTable512 (gps.c:145-213) are an
//   amplitude-250, 512-entry, half-sample-offset sine with sin[k] = sin[255-k],
//   sin[k+256] = -sin[k], cos[k] = sin[(k+128) mod 512]; the first quarter wave
//   below determines both (entry 35 is 105 where round(250 sin(..)) gives 106).
// G2 delays: ICD-GPS-200 code phase assignments for PRN 1..32 (gps.c:273-278).
#ifndef GPSIQ_TABLES_H
#define GPSIQ_TABLES_H

#include <stdint.h>

namespace gpsiq {

static const int16_t kQuarterWave[128] = {
      2,   5,   8,  11,  14,  17,  20,  23,  26,  29,  32,  35,  38,  41,  44,  47,
     50,  53,  56,  59,  62,  65,  68,  71,  74,  77,  80,  83,  86,  89,  91,  94,
     97, 100, 103, 105, 108, 111, 114, 116, 119, 122, 125, 127, 130, 132, 135, 138,
    140, 143, 145, 148, 150, 153, 155, 157, 160, 162, 164, 167, 169, 171, 173, 176,
    178, 180, 182, 184, 186, 188, 190, 192, 194, 196, 198, 200, 202, 204, 205, 207,
    209, 210, 212, 214, 215, 217, 218, 220, 221, 223, 224, 225, 227, 228, 229, 230,
    232, 233, 234, 235, 236, 237, 238, 239, 240, 241, 241, 242, 243, 244, 244, 245,
    245, 246, 247, 247, 248, 248, 248, 249, 249, 249, 249, 250, 250, 250, 250, 250,
};

static const uint16_t kG2Delay[32] = {
      5,   6,   7,   8,  17,  18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
    469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,
};

inline int sin512(int k)
{
    k &= 511;
    int h = k & 255;
    int v = kQuarterWave[h < 128 ? h : 255 - h];
    return k < 256 ? v : -v;
}

// Packed C/A code with wrap-around extension: bit j of word w is chip (32w + j) mod 1023,
// for 32w + j < 1023 + 65, so any 64-chip window starting at chip 0..1022 is contiguous.
constexpr int kPrnExtWords = 36;

// Device-resident constant tables (one copy per context).
struct DeviceTables {
    uint32_t prn_ext[32][kPrnExtWords];
    int16_t  quarter_wave[128];
};

}  // namespace gpsiq
#endif
