// nid_pose_estimation.cpp -- driver with the control flow of the reference's
// NID_pose_estimation.cpp (main: :55-399): read the config, load the pair and the
// ground truth, disturb the true pose, build the one-vertex / cells^2-edge graph,
// run 10 Levenberg-Marquardt iterations, print the 6-dof error and append it to
// nid_error.csv.  Host code only; every NID evaluation goes through the legacy
// operator signatures into the HIP library.
//
// Inputs: the config keys of config_eth_cvg.yaml (image0_id, image1_id,
// image0_type, image1_type, dataset, im_address, depth_factor, fx, fy, cx, cy,
// use_gpu) plus optional cell / bin_num / iterations / fused / resident / strict_math / pyramid_levels.  Images are read from
// <im_address><type>/<id>.png and <im_address>depth/<id>.png as in the reference (decoded
// by host/nid_png.cpp on zlib; colour -> grey with the reference's imread/cvtColor channel
// order), or the same stems with .pgm (binary PGM, 8-bit grey / 16-bit depth;
// tools/make_dataset.py writes either for the synthetic pair).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "g2o_min/g2o_min.h"
#include "../host/nid_pose_problem.h"

namespace {

std::string strip(const std::string &s) {
  size_t a = s.find_first_not_of(" \t\r\n'\"");
  size_t b = s.find_last_not_of(" \t\r\n'\"");
  if (a == std::string::npos) return "";
  return s.substr(a, b - a + 1);
}

// the flat "key: value" subset of OpenCV FileStorage YAML that config_eth_cvg.yaml uses
std::map<std::string, std::string> read_config(const std::string &path) {
  std::map<std::string, std::string> kv;
  std::ifstream f(path.c_str());
  std::string line;
  while (std::getline(f, line)) {
    const size_t h = line.find('#');
    if (h != std::string::npos) line = line.substr(0, h);
    if (line.empty() || line[0] == '%') continue;
    const size_t c = line.find(':');
    if (c == std::string::npos) continue;
    kv[strip(line.substr(0, c))] = strip(line.substr(c + 1));
  }
  return kv;
}

bool read_pgm(const std::string &path, int *rows, int *cols, int *maxval, std::vector<uint16_t> *out) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) return false;
  std::string magic;
  f >> magic;
  if (magic != "P5") return false;
  int vals[3], got = 0;
  while (got < 3) {
    f >> std::ws;
    if (f.peek() == '#') { std::string c; std::getline(f, c); continue; }
    if (!(f >> vals[got])) return false;
    got++;
  }
  f.get();  // single whitespace after maxval
  *cols = vals[0]; *rows = vals[1]; *maxval = vals[2];
  const size_t n = (size_t)vals[0] * vals[1];
  out->resize(n);
  if (*maxval < 256) {
    std::vector<uint8_t> b(n);
    f.read(reinterpret_cast<char *>(b.data()), (std::streamsize)n);
    for (size_t i = 0; i < n; i++) (*out)[i] = b[i];
  } else {
    std::vector<uint8_t> b(2 * n);
    f.read(reinterpret_cast<char *>(b.data()), (std::streamsize)(2 * n));
    for (size_t i = 0; i < n; i++) (*out)[i] = (uint16_t)((b[2 * i] << 8) | b[2 * i + 1]);  // PGM is big-endian
  }
  return (bool)f;
}

// <stem>.png (host/nid_png.cpp) if it exists, else <stem>.pgm; depth = 16-bit grey
bool read_image(const std::string &stem, bool depth, int *rows, int *cols, std::vector<uint16_t> *out) {
  const std::string png = stem + ".png";
  int r = 0, c = 0;
  if (nid_png_info(png.c_str(), &r, &c, nullptr, nullptr) == 0) {
    out->resize((size_t)r * c);
    *rows = r; *cols = c;
    if (depth) return nid_png_read_u16(png.c_str(), &r, &c, out->data(), out->size()) == 0;
    std::vector<uint8_t> g((size_t)r * c);
    if (nid_png_read_gray_u8(png.c_str(), 0, &r, &c, g.data(), g.size()) != 0) return false;
    for (size_t i = 0; i < g.size(); i++) (*out)[i] = g[i];
    return true;
  }
  int mv = 0;
  return read_pgm(stem + ".pgm", rows, cols, &mv, out);
}

// ReadGroundtruth, NID_pose_estimation.cpp:434-530: "ts tx ty tz qx qy qz qw", line index = frame id
std::vector<g2o::Matrix4d> read_groundtruth(const std::string &path) {
  std::vector<g2o::Matrix4d> all;
  std::ifstream f(path.c_str());
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    double ts, px, py, pz, qx, qy, qz, qw;
    if (!(ss >> ts >> px >> py >> pz >> qx >> qy >> qz >> qw)) continue;
    const g2o::Matrix3d R = g2o::Quaterniond(qw, qx, qy, qz).toRotationMatrix();
    g2o::Matrix4d T;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T(r, c) = R(r, c);
    T(0, 3) = px; T(1, 3) = py; T(2, 3) = pz; T(3, 3) = 1.0;
    all.push_back(T);
  }
  return all;
}

g2o::Matrix3d rot_axis(int axis, double a) {
  g2o::Matrix3d R = g2o::Matrix3d::Identity();
  const double c = std::cos(a), s = std::sin(a);
  const int i = (axis + 1) % 3, j = (axis + 2) % 3;
  R(i, i) = c; R(i, j) = -s; R(j, i) = s; R(j, j) = c;
  return R;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc != 2) {
    std::cout << "usage './nid_pose_estimation path_to_config.yaml', image1's timestamp should be smaller than image2"
              << std::endl;
    return 0;
  }
  std::map<std::string, std::string> fc = read_config(argv[1]);
  const std::string type0 = fc["image0_type"], type1 = fc["image1_type"], id0 = fc["image0_id"], id1 = fc["image1_id"];
  const std::string dataset = fc["dataset"], im_add = fc["im_address"];
  const double depth_factor = 1.0 / std::atoi(fc["depth_factor"].c_str());  // :73
  const double fx = std::atof(fc["fx"].c_str()), fy = std::atof(fc["fy"].c_str());
  const double cx = std::atof(fc["cx"].c_str()), cy = std::atof(fc["cy"].c_str());
  const bool use_gpu = fc.count("use_gpu") ? std::atoi(fc["use_gpu"].c_str()) != 0 : true;
  const int cell = fc.count("cell") ? std::atoi(fc["cell"].c_str()) : 16;        // :26
  const int bin_num = fc.count("bin_num") ? std::atoi(fc["bin_num"].c_str()) : 10;  // :27
  const int iterations = fc.count("iterations") ? std::atoi(fc["iterations"].c_str()) : 10;  // :340
  if (!use_gpu) {
    std::cerr << "use_gpu: 0 selects the reference's CPU edge, which this build does not contain (it exists only as "
                 "the test oracle).  Set use_gpu: 1.\n";
    return 2;
  }
  const int pose_id0 = std::atoi(id0.c_str()), pose_id1 = std::atoi(id1.c_str());
  std::cout << "optimize relative pose between " << pose_id0 << " and " << pose_id1 << std::endl;

  int rows = 0, cols = 0, r2 = 0, c2 = 0;
  std::vector<uint16_t> g0, g1, d0;
  // <id>.png as in the reference (:84-113; colour -> grey with its imread/cvtColor channel quirk), else <id>.pgm
  if (!read_image(im_add + type0 + "/" + id0, false, &rows, &cols, &g0) ||
      !read_image(im_add + type1 + "/" + id1, false, &r2, &c2, &g1) || r2 != rows || c2 != cols ||
      !read_image(im_add + "depth/" + id0, true, &r2, &c2, &d0) || r2 != rows || c2 != cols) {
    std::cerr << "cannot read the image pair / depth under " << im_add << std::endl;
    return 1;
  }
  std::cout << "image size [" << cols << " x " << rows << "]" << std::endl;
  std::vector<g2o::Matrix4d> gt = read_groundtruth(im_add + "groundtruth.txt");
  if ((int)gt.size() <= std::max(pose_id0, pose_id1)) {
    std::cerr << "cannot find the file that contains groundtruth" << std::endl;
    return 1;
  }
  const g2o::Matrix4d T_wc0 = gt[pose_id0], T_wc1 = gt[pose_id1];

  // T_cw1 = inverse(T_wc1); disturbance :186-212
  g2o::Matrix3d R_wc1;
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R_wc1(r, c) = T_wc1(r, c);
  g2o::Matrix3d r_cw1 = R_wc1.transpose();
  g2o::Vector3d t_wc1(T_wc1(0, 3), T_wc1(1, 3), T_wc1(2, 3));
  g2o::Vector3d t_cw1 = r_cw1 * t_wc1;
  for (int i = 0; i < 3; i++) t_cw1[i] = -t_cw1[i];
  const g2o::SE3Quat T_cw1_g2o(r_cw1, t_cw1);
  const g2o::Vector6d min_vec_gt = T_cw1_g2o.toMinimalVector();
  const double t_offset = 0.02, r_offset = 0.005;
  const g2o::Matrix3d rotation_dist = rot_axis(0, r_offset * M_PI) * rot_axis(1, r_offset * M_PI) * rot_axis(2, r_offset * M_PI);
  r_cw1 = rotation_dist * r_cw1;
  t_cw1 = t_cw1 + g2o::Vector3d(0.5 * t_offset, -t_offset, -t_offset);
  const g2o::SE3Quat start(r_cw1, t_cw1);
  const g2o::Vector6d error_ori = min_vec_gt - start.toMinimalVector();
  std::cout << "the error to be minimized is (6d minimal form) \n";
  for (int i = 0; i < 6; i++) std::cout << error_ori[i] << (i < 5 ? " " : "\n");

  std::vector<uint8_t> im0(g0.size()), im1(g1.size());
  for (size_t i = 0; i < g0.size(); i++) { im0[i] = (uint8_t)g0[i]; im1[i] = (uint8_t)g1[i]; }
  nid_pose_problem pb;
  std::memset(&pb, 0, sizeof(pb));
  pb.rows = rows; pb.cols = cols; pb.cell_num = cell; pb.bin_num = bin_num; pb.iterations = iterations;
  pb.fx = fx; pb.fy = fy; pb.cx = cx; pb.cy = cy; pb.depth_factor = depth_factor; pb.huber_delta = std::sqrt(0.95);
  pb.im0 = im0.data(); pb.im1 = im1.data(); pb.depth_u16 = d0.data(); pb.T_wc0_colmajor = T_wc0.data();
  pb.fused = fc.count("fused") ? std::atoi(fc["fused"].c_str()) : 0;
  pb.strict_math = fc.count("strict_math") ? std::atoi(fc["strict_math"].c_str()) : 0;
  // not in the reference: "resident: 1" answers the single-pose evaluations of the optimisation from a kernel that
  // stays on the device (nid_set_resident in include/nid/nid_c.h); best with "fused: 4"
  if (fc.count("resident")) nid_host_set_resident(std::atoi(fc["resident"].c_str()));
  if (fc.count("devices")) {
    // not in the reference: the cells of the pair sharded over several GPUs of this node (include/nid/nid_multi.h),
    // e.g. "devices: 0,1,2,3"; "reduce_rccl: 1" sums the 6x6 blocks with RCCL instead of on the host
    std::vector<int32_t> devs;
    std::stringstream ss(fc["devices"]);
    for (std::string tok; std::getline(ss, tok, ',');)
      if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
    if (devs.empty()) { std::cerr << "devices: expected a comma-separated list of GPU ids" << std::endl; return 1; }
    nid_host_set_devices(devs.data(), (int)devs.size(), fc.count("reduce_rccl") ? std::atoi(fc["reduce_rccl"].c_str()) : 0);
  }

  if (fc.count("mode") && fc["mode"] == "standard_property") {
    // the reference's second program (NID_standard_property.cpp): plain-histogram NID of every cell at the
    // ground-truth relative pose, "final nid is ..." on stdout
    pb.bin_num = fc.count("bin_num") ? bin_num : 8;  // :11
    double gt7[7], final_nid = 0.0;
    T_cw1_g2o.toPose7(gt7);
    std::vector<char> out(1 << 18);
    if (nid_host_standard_property(&pb, gt7, &final_nid, out.data(), (int)out.size()) != 0) {
      std::cerr << out.data();
      return 1;
    }
    std::cout << out.data();
    return 0;
  }

  double pose7[7];
  start.toPose7(pose7);
  std::vector<nid_host_lm_record> trace(iterations);
  std::vector<char> log(65536);
  std::cout << "enter optimization ............. 0" << std::endl;
  // optional coarse-to-fine schedule (not in the reference; host/nid_pyramid.cpp): pyramid_levels > 1
  const int levels = fc.count("pyramid_levels") ? std::atoi(fc["pyramid_levels"].c_str()) : 1;
  const int done = levels > 1
                       ? nid_host_run_pyramid_lm(&pb, levels, pose7, nullptr, 0, nullptr, log.data(), (int)log.size())
                       : nid_host_run_lm(&pb, pose7, trace.data(), iterations, log.data(), (int)log.size());
  std::cerr << log.data();
  if (done < 0) { std::cerr << "optimisation failed" << std::endl; return 1; }

  const g2o::SE3Quat recov = g2o::SE3Quat::fromPose7(pose7);
  const g2o::Vector6d error = min_vec_gt - recov.toMinimalVector();
  std::cout << "the final error is \n";
  for (int i = 0; i < 6; i++) std::cout << error[i] << (i < 5 ? " " : "\n");
  std::ofstream of("nid_error.csv", std::ofstream::out | std::ofstream::app);  // :67, :361
  of << error[0] << "," << error[1] << "," << error[2] << "," << error[3] << "," << error[4] << "," << error[5] << ","
     << pose_id0 << "," << pose_id1 << std::endl;
  const g2o::Matrix4d pose = recov.to_homogeneous_matrix();
  std::cout << "pose optimized \n";
  for (int r = 0; r < 4; r++) {
    for (int c = 0; c < 4; c++) std::cout << pose(r, c) << (c < 3 ? " " : "\n");
  }
  return 0;
}
