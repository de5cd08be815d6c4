// trajectory_main <in.bin> <n> <tum.txt> <kitti.txt>: n records of (16 f32 Tcw, f64 timestamp, i32 lost) through host/Trajectory.h
#include <cstdio>
#include <cstdlib>
#include "Trajectory.h"
int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    FILE* in = fopen(argv[1], "rb");
    if (!in) return 3;
    const int n = atoi(argv[2]);
    std::vector<sdfe::TrajectoryPose> poses(n);
    for (int i = 0; i < n; i++) {
        int lost = 0;
        if (fread(poses[i].Tcw, 4, 16, in) != 16 || fread(&poses[i].timestamp, 8, 1, in) != 1 || fread(&lost, 4, 1, in) != 1) return 4;
        poses[i].lost = lost != 0;
    }
    fclose(in);
    return sdfe::SaveTrajectoryTUM(argv[3], poses) && sdfe::SaveTrajectoryKITTI(argv[4], poses) ? 0 : 5;
}
