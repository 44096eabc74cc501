// tools/parse_experiment.cpp -- design data for a wave-per-block BGZF encoder (the write side): what a matcher
// that looks up G positions at once -- candidates from the table as the earlier groups left it, every position entered, optionally
// a check of the distances 1 .. near_d for runs and short periods, buckets of `ways` entries -- loses or gains against the serial
// greedy matcher (G = 1) on an uncompressed BAM stream.  Prints the compression ratio of the token stream under an ideal dynamic
// Huffman code (entropy of the symbol counts + extra bits + 86 bytes of framing per 0xFF00-byte block).
//   g++ -O2 -o parse_experiment parse_experiment.cpp;  ./parse_experiment stream.raw G hash_bits near_d [ways]
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
static inline uint32_t load32(const uint8_t* p){uint32_t v;memcpy(&v,p,4);return v;}
static uint32_t lensym(uint32_t len,uint32_t*eb){*eb=0;if(len==258)return 285;uint32_t l=len-3;if(l<8)return 257+l;uint32_t e=0;for(uint32_t t=l>>3;t;t>>=1)++e;*eb=e;uint32_t first=4u<<e;return 261+4*e+((l-first)>>e);}
static uint32_t distsym(uint32_t dist,uint32_t*eb){*eb=0;uint32_t d=dist-1;if(d<4)return d;uint32_t e=0;for(uint32_t t=d>>2;t;t>>=1)++e;*eb=e;uint32_t first=2u<<e;return 2*e+2+((d-first)>>e);}
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb");std::vector<uint8_t> d;uint8_t buf[65536];size_t k;while((k=fread(buf,1,sizeof buf,f))>0)d.insert(d.end(),buf,buf+k);fclose(f);
  const int G=atoi(argv[2]); const int hash_bits=atoi(argv[3]); const int near_d=atoi(argv[4]); const int ways=argc>5?atoi(argv[5]):1;
  double total_bits=0;uint64_t ntok=0;
  for(size_t off=0;off<d.size();off+=0xFF00){
    const uint8_t*in=d.data()+off;uint32_t n=d.size()-off<0xFF00?d.size()-off:0xFF00;
    std::vector<uint32_t> tab((size_t)ways<<hash_bits,0);
    std::vector<uint32_t> fl(286,0),fd(30,0);uint64_t extra=0;
    std::vector<uint32_t> best(n,0),bd(n,0);
    for(uint32_t g0=0;g0<n;g0+=G){
      uint32_t g1=g0+G<n?g0+G:n;
      // lookups against the table of earlier groups
      for(uint32_t p=g0;p<g1;++p){ if(p+4>n)continue; uint32_t x=load32(in+p); uint32_t h=(x*2654435761u)>>(32-hash_bits);
        uint32_t lim=n-p<258?n-p:258;
        for(int w=0;w<ways;++w){uint32_t c1=tab[(size_t)h*ways+w]; if(!c1)continue; uint32_t c=c1-1; if(p-c>32768||load32(in+c)!=x)continue; uint32_t l=4; while(l<lim&&in[c+l]==in[p+l])++l; if(l>best[p]){best[p]=l;bd[p]=p-c;}}
        for(int dd=1;dd<=near_d;++dd){ if(p<(uint32_t)dd)break; uint32_t c=p-dd; if(load32(in+c)!=x)continue; uint32_t l=4; while(l<lim&&in[c+l]==in[p+l])++l; if(l>best[p]){best[p]=l;bd[p]=dd;} }
      }
      // inserts: per hash the highest positions of the group (ways newest)
      for(uint32_t p=g0;p<g1;++p){ if(p+4>n)continue; uint32_t x=load32(in+p); uint32_t h=(x*2654435761u)>>(32-hash_bits); uint32_t*b=&tab[(size_t)h*ways]; for(int w=ways-1;w>0;--w)b[w]=b[w-1]; b[0]=p+1; }
    }
    uint32_t i=0; while(i<n){ if(best[i]>=4){uint32_t eb;fl[lensym(best[i],&eb)]++;extra+=eb;fd[distsym(bd[i],&eb)]++;extra+=eb;i+=best[i];} else {fl[in[i]]++;++i;} ++ntok; }
    fl[256]=1;
    auto ent=[&](std::vector<uint32_t>&v){double t=0,s=0;for(auto x:v)t+=x;for(auto x:v)if(x)s+=x*-log2(x/t);return s;};
    total_bits+=ent(fl)+ent(fd)+extra+26*8+60*8;
  }
  printf("G=%d hash_bits=%d near_d=%d ways=%d: ratio %.3f tokens %llu\n",G,hash_bits,near_d,ways,d.size()*8.0/total_bits,(unsigned long long)ntok);
}
