// TEST INFRASTRUCTURE — a minimal, functional stand-in for the ROS 1 / OpenCV / tf2 / cv_bridge / image_transport headers
// that include/aruco_slam/aruco_slam.h, include/aruco_slam/map_loader.h and the reference's own node source
// (src/aruco_slam_node.cpp, compiled in place in the build container) use.  Not a ROS implementation: one process, no
// transport.  `ros::spin()` replays a scripted scenario (directory named by $ASLAM_STUB_SCENARIO):
//   params.txt   "<key> <value>" lines of the parameter server; "tf <target> <source> tx ty tz qx qy qz qw" transforms
//   events.txt   "caminfo fx fy cx cy nD d0 .."  |  "enc <t> <wl> <wr>"  |  "img <t> <file> <rows> <cols> <channels>"
// delivering each event to the subscribed callback with ros::Time::now() == <t>, and every publish() is appended to
// out.txt as text so that a test can compare what the node published with the ctypes path.  Never shipped.
#pragma once
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------------- OpenCV core (cv::Mat)
#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
namespace cv {
template <class T> struct DepthOf;
template <> struct DepthOf<unsigned char> { enum { value = CV_8U }; };
template <> struct DepthOf<float> { enum { value = CV_32F }; };
template <> struct DepthOf<double> { enum { value = CV_64F }; };
class Mat {
public:
    int rows = 0, cols = 0;
    unsigned char* data = nullptr;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, void* ext, size_t st = 0) : rows(r), cols(c), data(static_cast<unsigned char*>(ext)), type_(type) {
        step = st ? st : static_cast<size_t>(c) * elemSize();
    }
    template <class T> explicit Mat(const std::vector<T>& v, bool copy = false) {           // n x 1 column, as OpenCV does
        create(static_cast<int>(v.size()), 1, DepthOf<T>::value);
        if (!v.empty()) std::memcpy(data, v.data(), v.size() * sizeof(T));
        (void)copy;
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    int type() const { return type_; }
    int depth() const { return type_ & 7; }
    int channels() const { return (type_ >> 3) + 1; }
    size_t elemSize() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return static_cast<size_t>(sz[depth()] * channels()); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int y = 0; y < rows; y++) std::memcpy(m.data + static_cast<size_t>(y) * m.step, data + static_cast<size_t>(y) * step, static_cast<size_t>(cols) * elemSize());
        return m;
    }
    template <class T> T& at(int i, int j = 0) { return *reinterpret_cast<T*>(data + static_cast<size_t>(i) * step + static_cast<size_t>(j) * sizeof(T)); }
    template <class T> const T& at(int i, int j = 0) const { return *reinterpret_cast<const T*>(data + static_cast<size_t>(i) * step + static_cast<size_t>(j) * sizeof(T)); }
private:
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        step = static_cast<size_t>(c) * elemSize();
        own_ = std::make_shared<std::vector<unsigned char>>(static_cast<size_t>(r) * step + 1, 0);
        data = own_->data();
    }
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> own_;
};
}  // namespace cv

// ---------------------------------------------------------------------------------------------------- ros core
namespace ros_stub {
struct World {
    double clock = 0.0;
    std::map<std::string, std::string> params;
    std::map<std::string, std::array<double, 7>> tf;
    std::string dir;
    std::ofstream out;
    std::function<void(double, double, double)> on_encoder;                                   // t, wl, wr
    std::function<void(double, const std::string&, int, int, int)> on_image;                  // t, file, rows, cols, channels
    std::vector<double> caminfo;
    static World& get() { static World w; return w; }
    void load() {
        const char* d = std::getenv("ASLAM_STUB_SCENARIO");
        if (!d) throw std::runtime_error("ASLAM_STUB_SCENARIO not set");
        dir = d;
        std::ifstream p(dir + "/params.txt");
        std::string line;
        while (std::getline(p, line)) {
            std::istringstream s(line);
            std::string k;
            if (!(s >> k)) continue;
            if (k == "tf") {
                std::string a, b; std::array<double, 7> v{};
                s >> a >> b;
                for (double& x : v) s >> x;
                tf[a + "<-" + b] = v;
            } else { std::string v; std::getline(s, v); size_t i = v.find_first_not_of(" \t"); params[k] = i == std::string::npos ? "" : v.substr(i); }
        }
        out.open(dir + "/out.txt");
        out.precision(17);
    }
};
inline void logf(const char* lvl, const char* fmt, ...) {
    if (!std::getenv("ASLAM_STUB_VERBOSE")) return;
    va_list ap; va_start(ap, fmt);
    std::fprintf(stderr, "[%s] ", lvl); std::vfprintf(stderr, fmt, ap); std::fprintf(stderr, "\n");
    va_end(ap);
}
}  // namespace ros_stub

#define ROS_INFO(...) ros_stub::logf("INFO", __VA_ARGS__)
#define ROS_WARN(...) ros_stub::logf("WARN", __VA_ARGS__)
#define ROS_ERROR(...) ros_stub::logf("ERROR", __VA_ARGS__)
#define ROS_INFO_STREAM(x) do { } while (0)
#define ROS_ERROR_STREAM(x) do { } while (0)
#define ROS_INFO_STREAM_ONCE(x) do { } while (0)

namespace ros {
struct Duration {
    double s;
    Duration(double sec = 0) : s(sec) {}
    double toSec() const { return s; }
    bool sleep() const { return true; }
};
struct Time {
    double s;
    Time(double sec = 0) : s(sec) {}
    static Time now() { return Time(ros_stub::World::get().clock); }
    double toSec() const { return s; }
};
inline void init(int&, char**, const std::string&) { ros_stub::World::get().load(); }
inline bool ok() { return true; }
inline void shutdown() { ros_stub::World::get().out.close(); }

template <class M> void stub_record(const std::string& topic, const M& m);                     // specialised per message type below

class Publisher {
public:
    Publisher() {}
    explicit Publisher(const std::string& t) : topic_(t) {}
    template <class M> void publish(const M& m) const { stub_record(topic_, m); }
private:
    std::string topic_;
};
class Subscriber {};

class NodeHandle {
public:
    template <class T> bool getParam(const std::string& key, T& v) const {
        auto& p = ros_stub::World::get().params;
        auto it = p.find(key);
        if (it == p.end()) return false;
        std::istringstream s(it->second);
        s >> v;
        return !s.fail();
    }
    bool getParam(const std::string& key, std::string& v) const {
        auto& p = ros_stub::World::get().params;
        auto it = p.find(key);
        if (it == p.end()) return false;
        v = it->second;
        return true;
    }
    template <class M> Publisher advertise(const std::string& topic, uint32_t, bool = false) { return Publisher(topic); }
    template <class M, class T> Subscriber subscribe(const std::string&, uint32_t, void (T::*fn)(const std::shared_ptr<M const>&), T* obj);
};
void spin();
}  // namespace ros

// ---------------------------------------------------------------------------------------------------- messages
namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
struct Float32 { float data = 0; };
struct Float32MultiArray { std::vector<float> data; typedef std::shared_ptr<Float32MultiArray const> ConstPtr; typedef std::shared_ptr<Float32MultiArray> Ptr; };
}  // namespace std_msgs
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
struct PoseArray { std_msgs::Header header; std::vector<Pose> poses; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct PoseWithCovarianceStamped { std_msgs::Header header; PoseWithCovariance pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
struct QuaternionStamped { std_msgs::Header header; Quaternion quaternion; };
}  // namespace geometry_msgs
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2 };
    std_msgs::Header header; std::string ns; int id = 0; int type = 0; int action = 0;
    geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; ros::Duration lifetime;
};
struct MarkerArray { std::vector<Marker> markers; };
}  // namespace visualization_msgs
namespace sensor_msgs {
struct Image {
    std_msgs::Header header; uint32_t height = 0, width = 0; std::string encoding; uint8_t is_bigendian = 0; uint32_t step = 0;
    std::vector<uint8_t> data;
};
typedef std::shared_ptr<Image> ImagePtr;
typedef std::shared_ptr<Image const> ImageConstPtr;
struct CameraInfo { std_msgs::Header header; uint32_t height = 0, width = 0; std::string distortion_model; std::vector<double> D; std::array<double, 9> K{}; };
typedef std::shared_ptr<CameraInfo const> CameraInfoConstPtr;
}  // namespace sensor_msgs

namespace ros {
template <> inline void stub_record(const std::string& topic, const geometry_msgs::PoseWithCovarianceStamped& m) {
    auto& o = ros_stub::World::get().out;
    o << "pose " << topic << " " << m.header.frame_id << " " << m.pose.pose.position.x << " " << m.pose.pose.position.y << " " << m.pose.pose.position.z
      << " " << m.pose.pose.orientation.x << " " << m.pose.pose.orientation.y << " " << m.pose.pose.orientation.z << " " << m.pose.pose.orientation.w;
    for (double c : m.pose.covariance) o << " " << c;
    o << "\n";
}
template <> inline void stub_record(const std::string& topic, const visualization_msgs::MarkerArray& a) {
    auto& o = ros_stub::World::get().out;
    o << "markers " << topic << " " << a.markers.size();
    for (const auto& k : a.markers)
        o << " | " << k.id << " " << k.header.frame_id << " " << k.type << " " << k.scale.x << " " << k.scale.y << " " << k.scale.z << " " << k.color.r << " "
          << k.color.g << " " << k.color.b << " " << k.color.a << " " << k.pose.position.x << " " << k.pose.position.y << " " << k.pose.position.z << " "
          << k.pose.orientation.x << " " << k.pose.orientation.y << " " << k.pose.orientation.z << " " << k.pose.orientation.w << " " << k.lifetime.toSec();
    o << "\n";
}
inline void stub_record_image(const std::string& topic, const sensor_msgs::Image& m) {
    unsigned long long sum = 1469598103934665603ull;                                          // FNV-1a over the pixels
    for (uint8_t b : m.data) { sum ^= b; sum *= 1099511628211ull; }
    ros_stub::World::get().out << "image " << topic << " " << m.height << " " << m.width << " " << m.encoding << " " << sum << "\n";
}
}  // namespace ros

// ---------------------------------------------------------------------------------------------------- cv_bridge / image_transport
namespace cv_bridge {
struct CvImage {
    std_msgs::Header header; std::string encoding; cv::Mat image;
    std::shared_ptr<const void> tracked;                                                      // keeps a shared message alive
    CvImage() {}
    CvImage(const std_msgs::Header& h, const std::string& e, const cv::Mat& i) : header(h), encoding(e), image(i) {}
    sensor_msgs::ImagePtr toImageMsg() const {
        auto m = std::make_shared<sensor_msgs::Image>();
        m->header = header; m->height = static_cast<uint32_t>(image.rows); m->width = static_cast<uint32_t>(image.cols); m->encoding = encoding;
        m->step = static_cast<uint32_t>(image.cols * image.elemSize());
        m->data.resize(static_cast<size_t>(image.rows) * m->step);
        for (int y = 0; y < image.rows; y++) std::memcpy(m->data.data() + static_cast<size_t>(y) * m->step, image.data + static_cast<size_t>(y) * image.step, m->step);
        return m;
    }
};
typedef std::shared_ptr<CvImage const> CvImageConstPtr;
inline CvImageConstPtr toCvShare(const sensor_msgs::ImageConstPtr& src, const std::string& encoding) {
    auto out = std::make_shared<CvImage>();
    out->header = src->header; out->encoding = encoding;
    const int sc = src->encoding == "mono8" ? 1 : 3, dc = encoding == "mono8" ? 1 : 3;
    if (sc == dc) {                                                                            // aliasing, like the real toCvShare
        out->image = cv::Mat(static_cast<int>(src->height), static_cast<int>(src->width), CV_MAKETYPE(CV_8U, sc),
                             const_cast<uint8_t*>(src->data.data()), src->step);
        out->tracked = src;
    } else {
        cv::Mat m(static_cast<int>(src->height), static_cast<int>(src->width), CV_MAKETYPE(CV_8U, dc));
        for (uint32_t y = 0; y < src->height; y++)
            for (uint32_t x = 0; x < src->width; x++) {
                const uint8_t* s = src->data.data() + static_cast<size_t>(y) * src->step + static_cast<size_t>(x) * sc;
                uint8_t* d = m.data + static_cast<size_t>(y) * m.step + static_cast<size_t>(x) * dc;
                if (dc == 3) { d[0] = d[1] = d[2] = s[0]; } else { d[0] = static_cast<uint8_t>((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + 8192) >> 14); }
            }
        out->image = m;
    }
    return out;
}
}  // namespace cv_bridge

namespace image_transport {
class Publisher {
public:
    Publisher() {}
    explicit Publisher(const std::string& t) : topic_(t) {}
    void publish(const sensor_msgs::ImageConstPtr& m) const { ros::stub_record_image(topic_, *m); }
    void publish(const sensor_msgs::Image& m) const { ros::stub_record_image(topic_, m); }
private:
    std::string topic_;
};
class CameraSubscriber {};
class ImageTransport {
public:
    explicit ImageTransport(const ros::NodeHandle&) {}
    Publisher advertise(const std::string& topic, uint32_t, bool = false) { return Publisher(topic); }
    template <class T>
    CameraSubscriber subscribeCamera(const std::string&, uint32_t, void (T::*fn)(const sensor_msgs::ImageConstPtr&, const sensor_msgs::CameraInfoConstPtr&), T* obj) {
        ros_stub::World::get().on_image = [fn, obj](double t, const std::string& file, int rows, int cols, int ch) {
            auto& w = ros_stub::World::get();
            auto img = std::make_shared<sensor_msgs::Image>();
            img->height = static_cast<uint32_t>(rows); img->width = static_cast<uint32_t>(cols); img->encoding = ch == 1 ? "mono8" : "bgr8";
            img->step = static_cast<uint32_t>(cols * ch);
            img->data.resize(static_cast<size_t>(rows) * img->step);
            std::ifstream f(w.dir + "/" + file, std::ios::binary);
            f.read(reinterpret_cast<char*>(img->data.data()), static_cast<std::streamsize>(img->data.size()));
            if (!f) throw std::runtime_error("short image file " + file);
            auto ci = std::make_shared<sensor_msgs::CameraInfo>();
            if (w.caminfo.size() >= 4) {                                                       // fx fy cx cy d0 d1 ...
                ci->K = {w.caminfo[0], 0, w.caminfo[2], 0, w.caminfo[1], w.caminfo[3], 0, 0, 1};
                ci->D.assign(w.caminfo.begin() + 4, w.caminfo.end());
            }
            img->header.stamp = ros::Time(t);
            (obj->*fn)(img, ci);
        };
        return CameraSubscriber();
    }
};
}  // namespace image_transport

// ---------------------------------------------------------------------------------------------------- tf2
namespace tf2 {
struct TransformException : public std::runtime_error { explicit TransformException(const std::string& s) : std::runtime_error(s) {} };
}
namespace tf2_ros {
class Buffer {
public:
    geometry_msgs::TransformStamped lookupTransform(const std::string& target, const std::string& source, const ros::Time&, const ros::Duration& = ros::Duration(0)) const {
        auto& tf = ros_stub::World::get().tf;
        auto it = tf.find(target + "<-" + source);
        if (it == tf.end()) throw tf2::TransformException("no transform " + target + " <- " + source);
        geometry_msgs::TransformStamped t;
        t.header.frame_id = target; t.child_frame_id = source;
        t.transform.translation.x = it->second[0]; t.transform.translation.y = it->second[1]; t.transform.translation.z = it->second[2];
        t.transform.rotation.x = it->second[3]; t.transform.rotation.y = it->second[4]; t.transform.rotation.z = it->second[5]; t.transform.rotation.w = it->second[6];
        return t;
    }
};
class TransformListener { public: explicit TransformListener(Buffer&) {} };
}  // namespace tf2_ros

// ---------------------------------------------------------------------------------------------------- subscribe / spin
namespace ros {
template <class M, class T> Subscriber NodeHandle::subscribe(const std::string&, uint32_t, void (T::*fn)(const std::shared_ptr<M const>&), T* obj) {
    static_assert(std::is_same<M, std_msgs::Float32MultiArray>::value, "the stub delivers only the encoder topic");
    ros_stub::World::get().on_encoder = [fn, obj](double, double wl, double wr) {
        auto m = std::make_shared<std_msgs::Float32MultiArray>();
        m->data = {static_cast<float>(wl), static_cast<float>(wr)};
        (obj->*fn)(m);
    };
    return Subscriber();
}
inline void spin() {
    auto& w = ros_stub::World::get();
    std::ifstream ev(w.dir + "/events.txt");
    std::string line;
    while (std::getline(ev, line)) {
        std::istringstream s(line);
        std::string k;
        if (!(s >> k)) continue;
        if (k == "caminfo") { w.caminfo.clear(); double v; while (s >> v) w.caminfo.push_back(v); if (w.caminfo.size() >= 5) w.caminfo.erase(w.caminfo.begin() + 4); }
        else if (k == "enc") { double t, a, b; s >> t >> a >> b; w.clock = t; if (w.on_encoder) w.on_encoder(t, a, b); }
        else if (k == "img") { double t; std::string f; int r, c, ch; s >> t >> f >> r >> c >> ch; w.clock = t; if (w.on_image) w.on_image(t, f, r, c, ch); }
    }
    w.out.flush();
}
}  // namespace ros
