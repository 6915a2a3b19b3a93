#include <cstdint>
#include <cstdio>
#include <random>
static uint64_t mix64(uint64_t z){ z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}
static uint32_t ref(uint64_t key,uint64_t t,uint32_t n,uint32_t j){ uint64_t u=mix64(key+t*0xD1342543DE82EF95ull+j); return (uint32_t)(((u>>32)*(uint64_t)n)>>32);}
static uint32_t fast(uint64_t key,uint64_t t,uint32_t n,uint32_t j){
    uint64_t z = key + t * 0xD1342543DE82EF95ull + j;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
    const uint32_t cl = 0x133111EBu, ch = 0x94D049BBu;
    const uint32_t yh = (uint32_t)(((uint64_t)zl * cl) >> 32) + zl * ch + zh * cl;
    const uint32_t uh = yh ^ (yh >> 31);
    return (uint32_t)(((uint64_t)uh * (uint64_t)n) >> 32);
}
int main(){ std::mt19937_64 r(1); long bad=0; for(long i=0;i<50000000;i++){ uint64_t k=r(),t=r()>>40; uint32_t n=1+(r()%50), j=r()%16; if(ref(k,t,n,j)!=fast(k,t,n,j)) bad++; } printf("bad %ld\n",bad); return bad!=0; }
