// Command-line front end of the five-edge skeleton (xd-tts_amd/csrc/edge_floor.hip, also in the library as
// xdtts_edge_floor_us): the latency floor of one persistent-decoder step.  Developer tool.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_edges5 tools/ubench_edges5.hip && ./ubench_edges5 [steps] [T] [tuned 0|1] [lazy] [first]
#include "../xd-tts_amd/csrc/edge_floor.hip"

int main(int argc, char **argv) {
  const int nsteps = argc > 1 ? atoi(argv[1]) : 633, T = argc > 2 ? atoi(argv[2]) : 100;
  xdtts_edge_floor::Delays dl = xdtts_edge_floor::kernel_delays(argc > 3 ? atoi(argv[3]) : 0);
  if (argc > 4) dl.lazy = atoi(argv[4]);
  if (argc > 5) dl.first = atoi(argv[5]);
  const double best = xdtts_edge_floor::measure(0, nsteps, T, dl, 5, true);
  if (best < 0) {
    printf("grid would not be co-resident, or an exchange failed\n");
    return 1;
  }
  printf("FLOOR_US_PER_STEP %.3f (%s poll delays)\n", best, dl.tuned ? "tuned" : "no");
  return 0;
}
