// ref_costfile_shim.cpp -- C entry points around VERBATIM SLICES of the reference's cost-file reader and writer.
//
// TEST INFRASTRUCTURE ONLY (see oracle/kbest_oracle.c for the rules).  This file holds no reference code.
// comparison.cpp as a whole needs Matplot++ / GTSAM (comparison.cpp:1-8), assignment.cpp Eigen / GTSAM / OpenCV, but the
// on-disk cost-matrix format lives in std-only lines.  oracle/Makefile cuts them out of the reference where it lies --
//     constsUtils.h:10,13-16           inf_d, file_exists
//     comparison.cpp:32-57             getCosts (the reader of generatedData/00/costMatrices/<id>_frame<N>.dat)
//     assignment.cpp:821-831           the body of saveAssignmentProb that writes the file (std::to_string per entry)
// -- into temporary files under /tmp (never into the repository) and hands their paths to this translation unit.
// The result, oracle/_ref/libref_costfile.so (git-ignored, a binary), records tests/golden/costfile_golden.npz and pins
// probabilisticsemslam_amd/costfile.py against the reference's own reader and writer.
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

#include REF_READER_SLICE

static void ref_write_body(const std::vector<double> &costMatrix, size_t nRows, size_t nCols, std::string savePath)
{
#include REF_WRITER_SLICE
}

extern "C" {

// getCosts reads "generatedData/00/costMatrices/<id>_frame<frame>.dat" relative to the working directory: `root` is made
// the working directory for the call.  Returns -1 if the file does not exist, -2 if `cap` is too small, else the number of
// values; out holds the rows as getCosts returns them ([row][col], row-major), *nRows / *nCols the shape.
int64_t ref_get_costs(const char *root, const char *id, int64_t frame, double *out, int64_t cap, int64_t *nRows, int64_t *nCols)
{
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd) || chdir(root) != 0) return -3;
    std::vector<std::vector<double>> costs;
    const bool ok = getCosts(std::string(id), (size_t)frame, costs);
    if (chdir(cwd) != 0) return -3;
    if (!ok) return -1;
    *nRows = (int64_t)costs.size();
    *nCols = costs.empty() ? 0 : (int64_t)costs[0].size();
    int64_t n = 0;
    for (const auto &row : costs) n += (int64_t)row.size();
    if (n > cap) return -2;
    int64_t o = 0;
    for (const auto &row : costs)
        for (double x : row) out[o++] = x;
    return n;
}

// the writer of saveAssignmentProb (assignment.cpp:821-831) on a column-major (nRows x nCols) matrix
void ref_write_costs(const double *costColMajor, int64_t nRows, int64_t nCols, const char *path)
{
    std::vector<double> c(costColMajor, costColMajor + (size_t)nRows * (size_t)nCols);
    ref_write_body(c, (size_t)nRows, (size_t)nCols, std::string(path));
}

}  // extern "C"
