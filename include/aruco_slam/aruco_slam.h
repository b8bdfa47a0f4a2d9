// aruco_slam/aruco_slam.h — drop-in replacement for the reference's header of the same name
// (gitAugust/Aruco_Slam include/aruco_slam/aruco_slam.h:40-60, 101-193) for ROS / OpenCV builds: the same
// `ArucoSlamIniteData` and the same public `ArucoSlam` surface, signature for signature,
//
//     ArucoSlam(const struct ArucoSlamIniteData&)                      aruco_slam.h:109
//     void addEncoder(const double& el, const double& er)              :116   (dt from ros::Time::now(), aruco_slam.cpp:24-32)
//     void addImage(const cv::Mat& img)                                :122
//     void setCameraParameters(const std::pair<cv::Mat, cv::Mat>&)     :129
//     visualization_msgs::MarkerArray toRosMappedMarkers()             :139
//     visualization_msgs::MarkerArray toRosDetectedMarkers()           :145
//     geometry_msgs::PoseWithCovarianceStamped toRosPose()             :151
//     cv::Mat getMarkedImg()                                           :152
//
// so that src/aruco_slam_node.cpp compiles UNCHANGED against this include directory and links with -laruco_slam_hip
// instead of the reference's libaruco_slam.  Everything is inline on top of the POD-only C-ABI (aruco_slam_hip.h): detection,
// pose and the EKF run in the HIP kernels; this header only converts cv::Mat / ROS types.  Eigen is not needed.
// tests/ros_stubs holds minimal stand-ins for the ROS / OpenCV headers so that the test-suite can compile and run this header
// (and, in the build container, the reference's own node source in place) without a ROS installation.
#ifndef ARUCO_SLAM_H
#define ARUCO_SLAM_H

#include <opencv2/core.hpp>
#include <geometry_msgs/PoseWithCovarianceStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <visualization_msgs/MarkerArray.h>
#include <ros/ros.h>

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../aruco_slam_hip.h"

#if defined(__has_include)
#if __has_include(<opencv2/aruco.hpp>)
#include <opencv2/aruco.hpp>
#define ARUCO_SLAM_HAVE_CV_ARUCO 1
#endif
#endif

#ifndef ARUCO_SLAM_MAX_ROWS            // largest frame the context is sized for (the reference has no such limit)
#define ARUCO_SLAM_MAX_ROWS 1080
#endif
#ifndef ARUCO_SLAM_MAX_COLS
#define ARUCO_SLAM_MAX_COLS 1920
#endif
#ifndef ARUCO_SLAM_MAX_LANDMARKS
#define ARUCO_SLAM_MAX_LANDMARKS 1000
#endif

/** Data loaded from parameters.yaml for the ArucoSlam class — field for field aruco_slam.h:40-60 */
struct ArucoSlamIniteData
{
    double Q_k, R_x, R_y, R_theta;
    double kl, kr, b;
    int markers_dictionary;
    double marker_length;
    std::string world_frame, camera_frame_optical, robot_frame_base;
    std::string image_topic_name, encoder_topic_name;
    std::string map_f;
    geometry_msgs::TransformStamped transformStamped_r2c;
    float USEFUL_DISTANCE_THRESHOLD = 3;
};

class ArucoSlam
{
public:
    ArucoSlam(const struct ArucoSlamIniteData &inite_data)                                   // aruco_slam.cpp:3-19
    {
        aslam_init init;
        aslam_default_init(&init);
        init.Q_k = inite_data.Q_k; init.R_x = inite_data.R_x; init.R_y = inite_data.R_y; init.R_theta = inite_data.R_theta;
        init.kl = inite_data.kl; init.kr = inite_data.kr; init.b = inite_data.b;
        init.marker_length = inite_data.marker_length;
        init.markers_dictionary = 16;                       // DICT_ARUCO_ORIGINAL is built in; any other table is handed over below
        init.useful_distance_threshold = inite_data.USEFUL_DISTANCE_THRESHOLD;
        const geometry_msgs::TransformStamped &t = inite_data.transformStamped_r2c;
        init.r2c_t[0] = t.transform.translation.x; init.r2c_t[1] = t.transform.translation.y; init.r2c_t[2] = t.transform.translation.z;
        init.r2c_q[0] = t.transform.rotation.x; init.r2c_q[1] = t.transform.rotation.y;
        init.r2c_q[2] = t.transform.rotation.z; init.r2c_q[3] = t.transform.rotation.w;
        if (init.r2c_q[0] == 0 && init.r2c_q[1] == 0 && init.r2c_q[2] == 0 && init.r2c_q[3] == 0)
            init.r2c_q[3] = 1;                              // default-constructed transform (TF lookup failed, aruco_slam_node.cpp:134-143)
        init.max_rows = ARUCO_SLAM_MAX_ROWS; init.max_cols = ARUCO_SLAM_MAX_COLS; init.max_batch = 1;
        init.max_landmarks = ARUCO_SLAM_MAX_LANDMARKS;
        init.max_updates_per_frame = 64;
        const int rc = aslam_create(&init, &ctx_);
        if (rc != ASLAM_OK)
            throw std::runtime_error("ArucoSlam: aslam_create failed (" + std::to_string(rc) + "): no usable gfx950 device?");
        if (inite_data.markers_dictionary != 16)
        {
#ifdef ARUCO_SLAM_HAVE_CV_ARUCO
            cv::Ptr<cv::aruco::Dictionary> d =
                cv::aruco::getPredefinedDictionary(static_cast<cv::aruco::PREDEFINED_DICTIONARY_NAME>(inite_data.markers_dictionary));
            check(aslam_set_dictionary_bytes(ctx_, d->markerSize, d->bytesList.rows, d->maxCorrectionBits, d->bytesList.data));
#else
            throw std::runtime_error("ArucoSlam: only DICT_ARUCO_ORIGINAL (16) is built in; other dictionaries need <opencv2/aruco.hpp>");
#endif
        }
    }
    ~ArucoSlam() { aslam_destroy(ctx_); }
    ArucoSlam(const ArucoSlam &) = delete;
    ArucoSlam &operator=(const ArucoSlam &) = delete;

    void addEncoder(const double &el, const double &er)                                      // aruco_slam.cpp:21-74
    {
        is_init_ = true;
        check(aslam_add_encoder(ctx_, el, er, ros::Time::now().toSec()));
    }

    void addImage(const cv::Mat &img)                                                        // aruco_slam.cpp:76-287
    {
        if (!is_init_)
            return;                                                                          // :84-85
        if (img.empty() || (img.channels() != 1 && img.channels() != 3) || img.depth() != CV_8U)
            throw std::runtime_error("ArucoSlam::addImage: 8-bit image with 1 or 3 channels expected");
        check(aslam_add_image(ctx_, img.data, img.rows, img.cols, img.channels(), static_cast<size_t>(img.step)));
        // markered_img_ = img.clone(); cv::aruco::drawDetectedMarkers(markered_img_, marker_corners, IDs);   :318-319
        markered_img_ = img.clone();
        if (markered_img_.channels() == 3)
            check(aslam_draw_detected_markers(ctx_, markered_img_.data, markered_img_.rows, markered_img_.cols,
                                              static_cast<size_t>(markered_img_.step)));
        fillMarkers(aslam_get_detected_markers, "base_link", detected_markers_);             // :336-347
        fillMarkers(aslam_get_map_markers, "world", detected_map_);                          // :265-281
    }

    void setCameraParameters(const std::pair<cv::Mat, cv::Mat> &cameraparameters)           // aruco_slam.h:129-133
    {
        const cv::Mat &Km = cameraparameters.first, &Dm = cameraparameters.second;
        if (Km.rows != 3 || Km.cols != 3)
            throw std::runtime_error("ArucoSlam::setCameraParameters: 3x3 camera matrix expected");
        double K[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) K[3 * i + j] = element(Km, i, j);
        std::vector<double> D;
        for (int i = 0; i < Dm.rows; i++)
            for (int j = 0; j < Dm.cols; j++) D.push_back(element(Dm, i, j));
        check(aslam_set_camera(ctx_, K, D.empty() ? nullptr : D.data(), static_cast<int>(D.size())));
    }

    visualization_msgs::MarkerArray toRosMappedMarkers() { return detected_map_; }
    visualization_msgs::MarkerArray toRosDetectedMarkers() { return detected_markers_; }

    geometry_msgs::PoseWithCovarianceStamped toRosPose()                                     // aruco_slam.cpp:378-410
    {
        aslam_pose_msg m;
        check(aslam_get_pose_msg(ctx_, &m));
        geometry_msgs::PoseWithCovarianceStamped rpose;
        rpose.header.frame_id = "world";
        rpose.pose.pose.position.x = m.position[0];
        rpose.pose.pose.position.y = m.position[1];
        rpose.pose.pose.position.z = m.position[2];
        rpose.pose.pose.orientation.x = m.orientation[0];
        rpose.pose.pose.orientation.y = m.orientation[1];
        rpose.pose.pose.orientation.z = m.orientation[2];
        rpose.pose.pose.orientation.w = m.orientation[3];
        for (int i = 0; i < 36; i++) rpose.pose.covariance[i] = m.covariance[i];
        return rpose;
    }

    cv::Mat getMarkedImg() { return markered_img_; }                                         // aruco_slam.h:152

    // ---- beyond the reference surface: the filter state (mu_ / sigma_ are private there, aruco_slam.h:157-158) ----
    aslam_ctx *handle() { return ctx_; }

private:
    typedef int (*marker_getter)(aslam_ctx *, int, int *, aslam_marker_msg *);
    void fillMarkers(marker_getter fn, const char *frame, visualization_msgs::MarkerArray &out)
    {
        int n = 0;
        check(fn(ctx_, 0, &n, nullptr));
        std::vector<aslam_marker_msg> v(static_cast<size_t>(n > 0 ? n : 1));
        if (n) check(fn(ctx_, n, &n, v.data()));
        out.markers.clear();
        for (int i = 0; i < n; i++)
        {                                                                                    // GenerateMarker, aruco_slam.cpp:289-305
            const aslam_marker_msg &m = v[static_cast<size_t>(i)];
            visualization_msgs::Marker k;
            k.id = m.id;
            k.header.frame_id = frame;
            k.type = visualization_msgs::Marker::CUBE;
            k.scale.x = m.scale[0]; k.scale.y = m.scale[1]; k.scale.z = m.scale[2];
            k.color.r = m.color[0]; k.color.g = m.color[1]; k.color.b = m.color[2]; k.color.a = m.color[3];
            k.pose.position.x = m.position[0]; k.pose.position.y = m.position[1]; k.pose.position.z = m.position[2];
            k.pose.orientation.x = m.orientation[0]; k.pose.orientation.y = m.orientation[1];
            k.pose.orientation.z = m.orientation[2]; k.pose.orientation.w = m.orientation[3];
            k.lifetime = ros::Duration(m.lifetime_sec);
            out.markers.push_back(k);
        }
    }
    static double element(const cv::Mat &m, int i, int j)
    {
        switch (m.depth())
        {
        case CV_64F: return m.at<double>(i, j);
        case CV_32F: return m.at<float>(i, j);
        default: throw std::runtime_error("ArucoSlam: CV_64F or CV_32F camera parameters expected");
        }
    }
    void check(int rc)
    {
        if (rc != ASLAM_OK) throw std::runtime_error(std::string("ArucoSlam: ") + aslam_last_error(ctx_));
    }

    aslam_ctx *ctx_ = nullptr;
    bool is_init_ = false;
    cv::Mat markered_img_;
    visualization_msgs::MarkerArray detected_map_;
    visualization_msgs::MarkerArray detected_markers_;
};

#endif
