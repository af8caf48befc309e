#include <stdio.h>
#include "../../kmcp_amd/csrc/dbformat.hpp"
int main(int argc, char** argv) {
  for (int i = 1; i < argc; i++) {
    std::string p = argv[i];
    if (p.size() > 4 && p.substr(p.size() - 4) == ".yml") { kmcpg::DbYml y; std::string e = kmcpg::read_db_yml(p, &y); (void)e; }
    else { kmcpg::UnikiHeader h; std::string e = kmcpg::read_uniki_header(p, &h); (void)e; }
  }
  return 0;
}
