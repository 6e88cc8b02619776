#include "hwang_amd.h"
#include "sampler.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include <algorithm>
static std::vector<uint8_t> rd(const char* p){FILE*f=fopen(p,"rb");fseek(f,0,SEEK_END);long n=ftell(f);fseek(f,0,SEEK_SET);std::vector<uint8_t> v(n);if(fread(v.data(),1,n,f)!=(size_t)n)exit(2);fclose(f);return v;}
int main(int argc,char**argv){
  auto avcc=rd("avcc.bin"); auto data=rd(argv[1]); auto offs=rd("offs.bin"); auto sizes=rd("sizes.bin");
  size_t n=offs.size()/8; int64_t out[8]; char err[256];
  int reps = argc>2?atoi(argv[2]):1; int gop = argc>3?atoi(argv[3]):60;
  // sparse: want the frame just before each keyframe-aligned GOP end -> B pictures are skipped
  size_t first = argc>4?atoi(argv[4]):gop-3;
  std::vector<uint64_t> wanted; for(size_t g=first; g<n; g+=gop) wanted.push_back(g);
  sampler_start();
  double best=1e9;
  for(int r=0;r<reps;r++){
   auto t0=std::chrono::steady_clock::now();
   int rc=hwang_h264_probe(avcc.data(),avcc.size(),data.data(),(const uint64_t*)offs.data(),(const uint64_t*)sizes.data(),n,wanted.data(),wanted.size(),out,err,sizeof err);
   if(rc){printf("err %s\n",err);return 1;}
   double dt=std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
   best=std::min(best,dt);
  }
  sampler_dump("samples.txt");
  printf("decoded %lld skipped %lld: best %.3fs -> %.3f ms per decoded picture\n",(long long)out[0],(long long)out[1],best,best*1e3/out[0]);
}
