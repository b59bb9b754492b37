#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cstring>
#include <cstdio>
#include <thread>
#include <vector>
#include <chrono>
#include <cstdlib>
using clk = std::chrono::steady_clock;
double ms(clk::time_point a){return std::chrono::duration<double,std::milli>(clk::now()-a).count();}
int main(int argc,char**argv){
  const char* path=argv[1]; int nt=atoi(argv[2]); int mode=atoi(argv[3]);
  int fd=open(path,O_RDONLY); struct stat sb; fstat(fd,&sb); size_t n=sb.st_size;
  for(int rep=0;rep<3;++rep){
    auto t0=clk::now();
    char* base; std::vector<char> buf;
    if(mode==0){ base=(char*)mmap(nullptr,n,PROT_READ,MAP_PRIVATE,fd,0);} 
    else if(mode==2){ base=(char*)mmap(nullptr,n,PROT_READ,MAP_PRIVATE|MAP_POPULATE,fd,0);} 
    else if(mode==3){ base=nullptr; } else { base=(char*)malloc(n); }
    double t_map=ms(t0);
    std::vector<std::thread> th; std::vector<size_t> cnt(nt);
    for(int t=0;t<nt;++t) th.emplace_back([&,t]{ size_t a=n*t/nt,z=n*(t+1)/nt; 
       if(mode==1){ size_t lo=a; while(lo<z){ ssize_t g=pread(fd,base+lo,z-lo,lo); if(g<=0)break; lo+=g;} }
       if(mode==3){ std::vector<char> b(1<<20); size_t c=0,lo=a; while(lo<z){ ssize_t g=pread(fd,b.data(),std::min<size_t>(b.size(),z-lo),lo); if(g<=0)break; const char*p=b.data();const char*e=p+g; while(p<e){const char*q=(const char*)memchr(p,'\n',e-p); if(!q)break; ++c; p=q+1;} lo+=g;} cnt[t]=c; return; }
       size_t c=0; const char*p=base+a; const char*e=base+z; while(p<e){ const char*q=(const char*)memchr(p,'\n',e-p); if(!q)break; ++c; p=q+1;} cnt[t]=c;});
    for(auto&x:th)x.join();
    double t_scan=ms(t0);
    if(mode==1) free(base); else if(mode!=3) munmap(base,n);
    size_t tot=0; for(auto c:cnt) tot+=c;
    printf("mode %d nt %d: map %.1f scan-done %.1f total %.1f ms lines %zu\n",mode,nt,t_map,t_scan,ms(t0),tot);
  }
}
