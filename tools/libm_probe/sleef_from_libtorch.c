#include <immintrin.h>
__m512 Sleef_expf16_u10(__m512);
__m512 Sleef_tanhf16_u10(__m512);
__m256 Sleef_expf8_u10avx2(__m256);
__m256 Sleef_tanhf8_u10avx2(__m256);
void sw_exp16(const float* in, float* out, long n) { for (long i = 0; i + 16 <= n; i += 16) _mm512_storeu_ps(out + i, Sleef_expf16_u10(_mm512_loadu_ps(in + i))); }
void sw_tanh16(const float* in, float* out, long n) { for (long i = 0; i + 16 <= n; i += 16) _mm512_storeu_ps(out + i, Sleef_tanhf16_u10(_mm512_loadu_ps(in + i))); }
void sw_exp8(const float* in, float* out, long n) { for (long i = 0; i + 8 <= n; i += 8) _mm256_storeu_ps(out + i, Sleef_expf8_u10avx2(_mm256_loadu_ps(in + i))); }
void sw_tanh8(const float* in, float* out, long n) { for (long i = 0; i + 8 <= n; i += 8) _mm256_storeu_ps(out + i, Sleef_tanhf8_u10avx2(_mm256_loadu_ps(in + i))); }
