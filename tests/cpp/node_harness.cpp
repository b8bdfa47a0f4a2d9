// TEST PROGRAM — drives include/aruco_slam/aruco_slam.h (the drop-in `ArucoSlam` class with the reference's signatures)
// the way a ROS node would, on the scripted single-process ROS stand-in of tests/ros_stubs: parameters -> ArucoSlamIniteData,
// camera_info -> setCameraParameters, image -> addImage + getMarkedImg + toRosDetectedMarkers + toRosMappedMarkers,
// encoder -> addEncoder + toRosPose, real_map from MapLoader.  Linked with -laruco_slam_hip (GPU box) or the CPU emulation
// build (container); tests/test_node_dropin.py compares everything it publishes with the ctypes path.
#include "aruco_slam/aruco_slam.h"
#include "aruco_slam/map_loader.h"
#include <cv_bridge/cv_bridge.h>
#include <image_transport/image_transport.h>
#include <std_msgs/Float32MultiArray.h>
#include <tf2_ros/transform_listener.h>

namespace {
struct Harness {
    ros::Publisher pose_pub, det_pub, map_pub, real_pub;
    image_transport::Publisher img_pub;
    std::unique_ptr<ArucoSlam> slam;
    bool have_camera = false;

    void onImage(const sensor_msgs::ImageConstPtr& msg, const sensor_msgs::CameraInfoConstPtr& info) {
        if (!have_camera) {
            cv::Mat K = cv::Mat::zeros(3, 3, CV_64F);
            for (int i = 0; i < 9; i++) K.at<double>(i / 3, i % 3) = info->K[static_cast<size_t>(i)];
            slam->setCameraParameters(std::make_pair(K, cv::Mat(info->D, true)));
            have_camera = true;
        }
        cv_bridge::CvImageConstPtr shared = cv_bridge::toCvShare(msg, "bgr8");
        slam->addImage(shared->image);
        img_pub.publish(cv_bridge::CvImage(std_msgs::Header(), "bgr8", slam->getMarkedImg()).toImageMsg());
        det_pub.publish(slam->toRosDetectedMarkers());
        map_pub.publish(slam->toRosMappedMarkers());
    }
    void onEncoder(const std_msgs::Float32MultiArray::ConstPtr& msg) {
        slam->addEncoder(msg->data.at(0), msg->data.at(1));
        pose_pub.publish(slam->toRosPose());
    }
};
}  // namespace

int main(int argc, char** argv) {
    ros::init(argc, argv, "aruco_slam_harness");
    ros::NodeHandle nh;
    image_transport::ImageTransport it(nh);
    tf2_ros::Buffer tf;
    tf2_ros::TransformListener listener(tf);
    Harness h;
    h.real_pub = nh.advertise<visualization_msgs::MarkerArray>("aruco_slam_node/real_map", 1, true);
    h.det_pub = nh.advertise<visualization_msgs::MarkerArray>("aruco_slam_node/detected_markers", 1);
    h.map_pub = nh.advertise<visualization_msgs::MarkerArray>("aruco_slam_node/detected_map", 1);
    h.pose_pub = nh.advertise<geometry_msgs::PoseWithCovarianceStamped>("aruco_slam_node/pose", 1);
    h.img_pub = it.advertise("aruco_slam_node/image", 1);

    std::string map_file;
    if (nh.getParam("/aruco_slam_node/map/map_file", map_file)) h.real_pub.publish(MapLoader(map_file).toRosRealMapMarkers());

    ArucoSlamIniteData d;
    const char* keys[] = {"odom/kl", "odom/kr", "odom/b", "covariance/Q_k", "covariance/R_x", "covariance/R_y", "covariance/R_theta", "aruco/marker_length"};
    double* dst[] = {&d.kl, &d.kr, &d.b, &d.Q_k, &d.R_x, &d.R_y, &d.R_theta, &d.marker_length};
    for (int i = 0; i < 8; i++) nh.getParam(std::string("/aruco_slam_node/") + keys[i], *dst[i]);
    nh.getParam("/aruco_slam_node/aruco/markers_dictionary", d.markers_dictionary);
    std::string base, cam;
    nh.getParam("/aruco_slam_node/frame/robot_frame_base", base);
    nh.getParam("/aruco_slam_node/frame/camera_frame_optical", cam);
    try { d.transformStamped_r2c = tf.lookupTransform(base, cam, ros::Time(0), ros::Duration(0.1)); }
    catch (tf2::TransformException& e) { ROS_WARN("%s", e.what()); }
    h.slam.reset(new ArucoSlam(d));

    std::string image_topic, encoder_topic;
    nh.getParam("/aruco_slam_node/topic/image", image_topic);
    nh.getParam("/aruco_slam_node/topic/encoder", encoder_topic);
    image_transport::CameraSubscriber cs = it.subscribeCamera(image_topic, 1, &Harness::onImage, &h);
    ros::Subscriber es = nh.subscribe(encoder_topic, 5, &Harness::onEncoder, &h);
    (void)cs; (void)es;
    ros::spin();
    ros::shutdown();
    return 0;
}
