// Stand-in for <voxblox_ros/ros_params.h>: the three loaders SemanticTsdfServer's first constructor calls.
#pragma once
#include <voxblox_ros/tsdf_server.h>
namespace voxblox {
inline TsdfMap::Config getTsdfMapConfigFromRosParam(const ros::NodeHandle&) { return TsdfMap::Config(); }
inline TsdfIntegratorBase::Config getTsdfIntegratorConfigFromRosParam(const ros::NodeHandle&) { return TsdfIntegratorBase::Config(); }
inline MeshIntegratorConfig getMeshIntegratorConfigFromRosParam(const ros::NodeHandle&) { return MeshIntegratorConfig(); }
}  // namespace voxblox
