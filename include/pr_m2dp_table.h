/* pr_m2dp_table.h — frozen M2DP plane-normal table (DATA, shared by product, oracle and tests).
 *
 * The reference builds the 64 plane normals with FLOAT trig (M2DP/M2DP.cpp:9-15):
 *   azm = (float)(-pi/2 + p*pi/4), elv = (float)(q*pi/32),
 *   n = (cosf(elv)*cosf(azm), cosf(elv)*sinf(azm), sinf(elv))   [float products, then widened]
 * cosf/sinf are libm-dependent in the last ulp, so the table is frozen once here
 * as IEEE-754 bit patterns generated with glibc 2.35 / g++ 11.4 (recipe: oracle/gen_m2dp_table.cpp).
 * Index k = p*16 + q.  xProj / yProj are derived from it in fp64 by every consumer (M2DP.cpp:17-25).
 */
#ifndef PR_M2DP_TABLE_H
#define PR_M2DP_TABLE_H
#include <stdint.h>
static const uint32_t PR_M2DP_VECN_BITS[64][3] = {
  {0xb33bbd2eu, 0xbf800000u, 0x00000000u}, /* p0 q0 */
  {0xb33ad5c0u, 0xbf7ec46du, 0x3dc8bd36u}, /* p0 q1 */
  {0xb33821b2u, 0xbf7b14beu, 0x3e47c5c2u}, /* p0 q2 */
  {0xb333a7afu, 0xbf74fa0bu, 0x3e94a032u}, /* p0 q3 */
  {0xb32d72bdu, 0xbf6c835eu, 0x3ec3ef16u}, /* p0 q4 */
  {0xb325922du, 0xbf61c598u, 0x3ef15ae9u}, /* p0 q5 */
  {0xb31c1969u, 0xbf54db31u, 0x3f0e39dau}, /* p0 q6 */
  {0xb3111fccu, 0xbf45e404u, 0x3f226799u}, /* p0 q7 */
  {0xb304c063u, 0xbf3504f3u, 0x3f3504f3u}, /* p0 q8 */
  {0xb2ee3361u, 0xbf226799u, 0x3f45e403u}, /* p0 q9 */
  {0xb2d09ab9u, 0xbf0e39dau, 0x3f54db31u}, /* p0 q10 */
  {0xb2b0ffc5u, 0xbef15aebu, 0x3f61c597u}, /* p0 q11 */
  {0xb28fb06fu, 0xbec3ef15u, 0x3f6c835eu}, /* p0 q12 */
  {0xb259fdb0u, 0xbe94a030u, 0x3f74fa0bu}, /* p0 q13 */
  {0xb2128117u, 0xbe47c5c4u, 0x3f7b14beu}, /* p0 q14 */
  {0xb193368du, 0xbdc8bd35u, 0x3f7ec46du}, /* p0 q15 */
  {0x3f3504f3u, 0xbf3504f3u, 0x00000000u}, /* p1 q0 */
  {0x3f3425ceu, 0xbf3425ceu, 0x3dc8bd36u}, /* p1 q1 */
  {0x3f318a85u, 0xbf318a85u, 0x3e47c5c2u}, /* p1 q2 */
  {0x3f2d3986u, 0xbf2d3986u, 0x3e94a032u}, /* p1 q3 */
  {0x3f273d74u, 0xbf273d74u, 0x3ec3ef16u}, /* p1 q4 */
  {0x3f1fa512u, 0xbf1fa512u, 0x3ef15ae9u}, /* p1 q5 */
  {0x3f168317u, 0xbf168317u, 0x3f0e39dau}, /* p1 q6 */
  {0x3f0bee0au, 0xbf0bee0au, 0x3f226799u}, /* p1 q7 */
  {0x3effffffu, 0xbeffffffu, 0x3f3504f3u}, /* p1 q8 */
  {0x3ee5acc6u, 0xbee5acc6u, 0x3f45e403u}, /* p1 q9 */
  {0x3ec9234eu, 0xbec9234eu, 0x3f54db31u}, /* p1 q10 */
  {0x3eaaa9f3u, 0xbeaaa9f3u, 0x3f61c597u}, /* p1 q11 */
  {0x3e8a8bd4u, 0xbe8a8bd4u, 0x3f6c835eu}, /* p1 q12 */
  {0x3e523043u, 0xbe523043u, 0x3f74fa0bu}, /* p1 q13 */
  {0x3e0d42b0u, 0xbe0d42b0u, 0x3f7b14beu}, /* p1 q14 */
  {0x3d8df1a8u, 0xbd8df1a8u, 0x3f7ec46du}, /* p1 q15 */
  {0x3f800000u, 0x00000000u, 0x00000000u}, /* p2 q0 */
  {0x3f7ec46du, 0x00000000u, 0x3dc8bd36u}, /* p2 q1 */
  {0x3f7b14beu, 0x00000000u, 0x3e47c5c2u}, /* p2 q2 */
  {0x3f74fa0bu, 0x00000000u, 0x3e94a032u}, /* p2 q3 */
  {0x3f6c835eu, 0x00000000u, 0x3ec3ef16u}, /* p2 q4 */
  {0x3f61c598u, 0x00000000u, 0x3ef15ae9u}, /* p2 q5 */
  {0x3f54db31u, 0x00000000u, 0x3f0e39dau}, /* p2 q6 */
  {0x3f45e404u, 0x00000000u, 0x3f226799u}, /* p2 q7 */
  {0x3f3504f3u, 0x00000000u, 0x3f3504f3u}, /* p2 q8 */
  {0x3f226799u, 0x00000000u, 0x3f45e403u}, /* p2 q9 */
  {0x3f0e39dau, 0x00000000u, 0x3f54db31u}, /* p2 q10 */
  {0x3ef15aebu, 0x00000000u, 0x3f61c597u}, /* p2 q11 */
  {0x3ec3ef15u, 0x00000000u, 0x3f6c835eu}, /* p2 q12 */
  {0x3e94a030u, 0x00000000u, 0x3f74fa0bu}, /* p2 q13 */
  {0x3e47c5c4u, 0x00000000u, 0x3f7b14beu}, /* p2 q14 */
  {0x3dc8bd35u, 0x00000000u, 0x3f7ec46du}, /* p2 q15 */
  {0x3f3504f3u, 0x3f3504f3u, 0x00000000u}, /* p3 q0 */
  {0x3f3425ceu, 0x3f3425ceu, 0x3dc8bd36u}, /* p3 q1 */
  {0x3f318a85u, 0x3f318a85u, 0x3e47c5c2u}, /* p3 q2 */
  {0x3f2d3986u, 0x3f2d3986u, 0x3e94a032u}, /* p3 q3 */
  {0x3f273d74u, 0x3f273d74u, 0x3ec3ef16u}, /* p3 q4 */
  {0x3f1fa512u, 0x3f1fa512u, 0x3ef15ae9u}, /* p3 q5 */
  {0x3f168317u, 0x3f168317u, 0x3f0e39dau}, /* p3 q6 */
  {0x3f0bee0au, 0x3f0bee0au, 0x3f226799u}, /* p3 q7 */
  {0x3effffffu, 0x3effffffu, 0x3f3504f3u}, /* p3 q8 */
  {0x3ee5acc6u, 0x3ee5acc6u, 0x3f45e403u}, /* p3 q9 */
  {0x3ec9234eu, 0x3ec9234eu, 0x3f54db31u}, /* p3 q10 */
  {0x3eaaa9f3u, 0x3eaaa9f3u, 0x3f61c597u}, /* p3 q11 */
  {0x3e8a8bd4u, 0x3e8a8bd4u, 0x3f6c835eu}, /* p3 q12 */
  {0x3e523043u, 0x3e523043u, 0x3f74fa0bu}, /* p3 q13 */
  {0x3e0d42b0u, 0x3e0d42b0u, 0x3f7b14beu}, /* p3 q14 */
  {0x3d8df1a8u, 0x3d8df1a8u, 0x3f7ec46du}, /* p3 q15 */
};
#endif
