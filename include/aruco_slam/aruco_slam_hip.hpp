// aruco_slam_hip.hpp — header-only C++ adapter with the public surface of the reference's `ArucoSlam`
// (include/aruco_slam/aruco_slam.h:101-193) on top of the POD-only C-ABI (include/aruco_slam_hip.h).
//
// The POD overloads compile anywhere (this image has neither OpenCV, Eigen nor ROS).  Where <opencv2/core.hpp> is
// available the `cv::Mat` overloads with the reference's exact signatures are enabled.  The visualisation getters
// (toRosPose / toRosDetectedMarkers / toRosMappedMarkers, aruco_slam.cpp:265-305,378-410) return the message CONTENT as
// plain structs (aslam_pose_msg / aslam_marker_msg); where the ROS message headers exist, the overloads at the bottom copy
// them field by field into geometry_msgs / visualization_msgs types.
#pragma once
#include "../aruco_slam_hip.h"
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#define ARUCO_SLAM_HIP_HAVE_OPENCV 1
#endif
#endif

namespace aruco_slam_hip {

// struct ArucoSlamIniteData (aruco_slam.h:40-60) without the ROS members; r2c carries transformStamped_r2c
struct ArucoSlamIniteData {
    double Q_k = 0.01, R_x = 100, R_y = 100, R_theta = 10;      // parameters.yaml:5-8
    double kl = 0.05, kr = 0.05, b = 0.09;                       // parameters.yaml:11-13
    int markers_dictionary = 16;                                 // parameters.yaml:16
    double marker_length = 0.27;                                 // parameters.yaml:17
    double r2c_translation[3] = {0, 0, 0};
    double r2c_rotation_xyzw[4] = {0, 0, 0, 1};
    float USEFUL_DISTANCE_THRESHOLD = 3;                         // aruco_slam.h:58
    int device_id = 0, max_landmarks = 256, max_rows = 720, max_cols = 1280;
};

struct Detection { int id; float corners[8]; double rvec[3], tvec[3]; };

class ArucoSlam {
public:
    explicit ArucoSlam(const ArucoSlamIniteData& d) {            // aruco_slam.h:109
        aslam_init init;
        aslam_default_init(&init);
        init.Q_k = d.Q_k; init.R_x = d.R_x; init.R_y = d.R_y; init.R_theta = d.R_theta;
        init.kl = d.kl; init.kr = d.kr; init.b = d.b;
        init.marker_length = d.marker_length; init.markers_dictionary = d.markers_dictionary;
        init.useful_distance_threshold = d.USEFUL_DISTANCE_THRESHOLD;
        for (int i = 0; i < 3; i++) init.r2c_t[i] = d.r2c_translation[i];
        for (int i = 0; i < 4; i++) init.r2c_q[i] = d.r2c_rotation_xyzw[i];
        init.device_id = d.device_id; init.max_landmarks = d.max_landmarks;
        init.max_rows = d.max_rows; init.max_cols = d.max_cols; init.max_batch = 1;
        int rc = aslam_create(&init, &ctx_);
        if (rc != ASLAM_OK) throw std::runtime_error("aslam_create failed (" + std::to_string(rc) + "): no usable gfx950 device?");
    }
    ~ArucoSlam() { aslam_destroy(ctx_); }
    ArucoSlam(const ArucoSlam&) = delete;
    ArucoSlam& operator=(const ArucoSlam&) = delete;

    // void addEncoder(const double& el, const double& er) — aruco_slam.h:116; `now_sec` = ros::Time::now().toSec()
    void addEncoder(const double& el, const double& er, double now_sec) { check(aslam_add_encoder(ctx_, el, er, now_sec)); }
    // void addImage(const cv::Mat& img) — aruco_slam.h:122; POD form: bgr8 (channels 3) or gray (1), borrowed for the call
    void addImage(const unsigned char* px, int rows, int cols, int channels, size_t step) {
        check(aslam_add_image(ctx_, px, rows, cols, channels, step));
    }
    // void setCameraParameters(const std::pair<cv::Mat, cv::Mat>&) — aruco_slam.h:129-133; K row-major 3x3
    void setCameraParameters(const double K[9], const double* D, int nD) { check(aslam_set_camera(ctx_, K, D, nD)); }

#ifdef ARUCO_SLAM_HIP_HAVE_OPENCV
    void addImage(const cv::Mat& img) { addImage(img.data, img.rows, img.cols, img.channels(), img.step[0]); }
    void setCameraParameters(const std::pair<cv::Mat, cv::Mat>& p) {
        cv::Mat K, D;
        p.first.convertTo(K, CV_64F);
        p.second.reshape(1, 1).convertTo(D, CV_64F);
        K = K.clone();
        setCameraParameters(K.ptr<double>(), D.total() ? D.ptr<double>() : nullptr, (int)D.total());
    }
#endif

    // mu_ / sigma_ (aruco_slam.h:182-183); sigma column-major N x N like Eigen::MatrixXd
    void state(std::vector<double>& mu, std::vector<double>& sigma) {
        int N = 0;
        check(aslam_get_state(ctx_, &N, nullptr, nullptr));
        mu.resize(N);
        sigma.resize((size_t)N * N);
        check(aslam_get_state(ctx_, &N, mu.data(), sigma.data()));
    }
    // what toRosPose() scatters into the 6x6 covariance (aruco_slam.cpp:399-407): x, y, theta and the 3x3 pose block
    void pose(double xyt[3], double cov3x3_rowmajor[9]) {
        std::vector<double> mu, s;
        state(mu, s);
        const size_t N = mu.size();
        for (int i = 0; i < 3; i++) { xyt[i] = mu[i]; for (int j = 0; j < 3; j++) cov3x3_rowmajor[i * 3 + j] = s[(size_t)j * N + i]; }
    }
    std::vector<Detection> detections() {                        // marker_corners / IDs / rvs / tvs (aruco_slam.cpp:309-314)
        int M = 0;
        check(aslam_get_detections(ctx_, &M, nullptr, nullptr, nullptr, nullptr));
        std::vector<int> ids(M); std::vector<float> c((size_t)M * 8); std::vector<double> r((size_t)M * 3), t((size_t)M * 3);
        check(aslam_get_detections(ctx_, &M, ids.data(), c.data(), r.data(), t.data()));
        std::vector<Detection> out(M);
        for (int i = 0; i < M; i++) {
            out[i].id = ids[i];
            for (int k = 0; k < 8; k++) out[i].corners[k] = c[(size_t)i * 8 + k];
            for (int k = 0; k < 3; k++) { out[i].rvec[k] = r[(size_t)i * 3 + k]; out[i].tvec[k] = t[(size_t)i * 3 + k]; }
        }
        return out;
    }
    // geometry_msgs::PoseWithCovarianceStamped toRosPose() — aruco_slam.h:151: frame "world", content as plain data
    aslam_pose_msg toRosPoseData() { aslam_pose_msg m; check(aslam_get_pose_msg(ctx_, &m)); return m; }
    // visualization_msgs::MarkerArray toRosMappedMarkers() — aruco_slam.h:145 (detected_map_, frame "world")
    std::vector<aslam_marker_msg> toRosMappedMarkersData() { return markers(aslam_get_map_markers); }
    // visualization_msgs::MarkerArray toRosDetectedMarkers() — aruco_slam.h:139 (detected_markers_, frame "base_link")
    std::vector<aslam_marker_msg> toRosDetectedMarkersData() { return markers(aslam_get_detected_markers); }
    // cv::aruco::getPredefinedDictionary stand-ins and DetectorParameters (aruco_slam.cpp:11-12, 313)
    void setDictionary(int markerSize, int nMarkers, int maxCorrectionBits, const unsigned char* bytesList) {
        check(aslam_set_dictionary_bytes(ctx_, markerSize, nMarkers, maxCorrectionBits, bytesList));
    }
    void setDetectorParameters(const aslam_detector_params& p) { check(aslam_set_detector_params(ctx_, &p)); }

    std::vector<int> landmarkIds() {                              // aruco_id_map (aruco_slam.h:164), by landmark index
        int L = 0;
        check(aslam_get_landmark_ids(ctx_, &L, nullptr));
        std::vector<int> ids(L);
        if (L) check(aslam_get_landmark_ids(ctx_, &L, ids.data()));
        return ids;
    }
    aslam_ctx* handle() { return ctx_; }

private:
    std::vector<aslam_marker_msg> markers(int (*fn)(aslam_ctx*, int, int*, aslam_marker_msg*)) {
        int n = 0;
        check(fn(ctx_, 0, &n, nullptr));
        std::vector<aslam_marker_msg> out(n);
        if (n) check(fn(ctx_, n, &n, out.data()));
        out.resize(n);
        return out;
    }
    void check(int rc) { if (rc != ASLAM_OK) throw std::runtime_error(std::string("aruco_slam_hip: ") + aslam_last_error(ctx_)); }
    aslam_ctx* ctx_ = nullptr;
};

#if defined(__has_include)
#if __has_include(<geometry_msgs/PoseWithCovarianceStamped.h>) && __has_include(<visualization_msgs/MarkerArray.h>)
} // namespace aruco_slam_hip
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <visualization_msgs/MarkerArray.h>
#include <ros/duration.h>
namespace aruco_slam_hip {
inline geometry_msgs::PoseWithCovarianceStamped toRosPose(ArucoSlam& s) {                      // aruco_slam.cpp:378-410
    const aslam_pose_msg m = s.toRosPoseData();
    geometry_msgs::PoseWithCovarianceStamped r;
    r.header.frame_id = "world";
    r.pose.pose.position.x = m.position[0]; r.pose.pose.position.y = m.position[1]; r.pose.pose.position.z = m.position[2];
    r.pose.pose.orientation.x = m.orientation[0]; r.pose.pose.orientation.y = m.orientation[1];
    r.pose.pose.orientation.z = m.orientation[2]; r.pose.pose.orientation.w = m.orientation[3];
    for (int i = 0; i < 36; i++) r.pose.covariance[i] = m.covariance[i];
    return r;
}
inline visualization_msgs::MarkerArray toRosMarkers(const std::vector<aslam_marker_msg>& v, const char* frame) {   // GenerateMarker, :289-305
    visualization_msgs::MarkerArray a;
    for (const aslam_marker_msg& m : v) {
        visualization_msgs::Marker k;
        k.id = m.id; k.header.frame_id = frame; k.type = visualization_msgs::Marker::CUBE;
        k.scale.x = m.scale[0]; k.scale.y = m.scale[1]; k.scale.z = m.scale[2];
        k.color.r = m.color[0]; k.color.g = m.color[1]; k.color.b = m.color[2]; k.color.a = m.color[3];
        k.pose.position.x = m.position[0]; k.pose.position.y = m.position[1]; k.pose.position.z = m.position[2];
        k.pose.orientation.x = m.orientation[0]; k.pose.orientation.y = m.orientation[1];
        k.pose.orientation.z = m.orientation[2]; k.pose.orientation.w = m.orientation[3];
        k.lifetime = ros::Duration(m.lifetime_sec);
        a.markers.push_back(k);
    }
    return a;
}
inline visualization_msgs::MarkerArray toRosMappedMarkers(ArucoSlam& s) { return toRosMarkers(s.toRosMappedMarkersData(), "world"); }
inline visualization_msgs::MarkerArray toRosDetectedMarkers(ArucoSlam& s) { return toRosMarkers(s.toRosDetectedMarkersData(), "base_link"); }
#endif
#endif

} // namespace aruco_slam_hip
