// Stand-in for <ros/ros.h> (ROS is not in this image): just enough of ros::NodeHandle for the reference's
// kimera_semantics_ros sources to be COMPILED by integration/check_server_patch.sh.  Nothing links against it.
#pragma once
#include <string>
namespace ros {
class NodeHandle {
 public:
  template <typename T>
  bool param(const std::string&, T& value, const T& fallback) const { value = fallback; return false; }
  template <typename T>
  bool getParam(const std::string&, T&) const { return false; }
};
}  // namespace ros
