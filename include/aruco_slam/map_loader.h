// aruco_slam/map_loader.h — drop-in for the reference's MapLoader (include/aruco_slam/map_loader.h:30-57,
// src/map_loader.cpp:7-118): the ground-truth map file behind the latched `real_map` topic, parsed by
// aslam_load_map_txt (same line format and rules, crossed roll / yaw fallbacks included) and wrapped into the
// visualization_msgs::MarkerArray the node publishes (aruco_slam_node.cpp:61-67).
#ifndef MAP_LOADER_H
#define MAP_LOADER_H

#include <string>
#include <vector>

#include <ros/ros.h>
#include <visualization_msgs/MarkerArray.h>

#include "../aruco_slam_hip.h"

using namespace std;          // the reference's header does this, and aruco_slam_node.cpp relies on nothing less

class MapLoader
{
public:
    MapLoader() = delete;
    MapLoader(const string &file_path) { loadMap(file_path); }
    visualization_msgs::MarkerArray toRosRealMapMarkers() { return real_map_; }

private:
    visualization_msgs::MarkerArray real_map_;
    void loadMap(const string &file_path)
    {
        int n = 0;
        if (aslam_load_map_txt(nullptr, file_path.c_str(), 0, &n, nullptr) != ASLAM_OK)
        {
            ROS_ERROR("Unable to open map file: %s", file_path.c_str());                     // map_loader.cpp:13-17
            return;
        }
        std::vector<aslam_marker_msg> v(static_cast<size_t>(n > 0 ? n : 1));
        if (n) aslam_load_map_txt(nullptr, file_path.c_str(), n, &n, v.data());
        for (int i = 0; i < n; i++)
        {                                                                                    // MapLoader::generateMarker, :96-118
            const aslam_marker_msg &m = v[static_cast<size_t>(i)];
            visualization_msgs::Marker k;
            k.id = m.id;
            k.header.frame_id = "world";
            k.type = visualization_msgs::Marker::CUBE;
            k.scale.x = m.scale[0]; k.scale.y = m.scale[1]; k.scale.z = m.scale[2];
            k.color.r = m.color[0]; k.color.g = m.color[1]; k.color.b = m.color[2]; k.color.a = m.color[3];
            k.pose.position.x = m.position[0]; k.pose.position.y = m.position[1]; k.pose.position.z = m.position[2];
            k.pose.orientation.x = m.orientation[0]; k.pose.orientation.y = m.orientation[1];
            k.pose.orientation.z = m.orientation[2]; k.pose.orientation.w = m.orientation[3];
            k.lifetime = ros::Duration(0);
            real_map_.markers.push_back(k);
        }
    }
};
#endif
